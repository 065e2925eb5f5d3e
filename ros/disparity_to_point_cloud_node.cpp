// ros/disparity_to_point_cloud_node.cpp -- the ROS node, source only (this
// image has no ROS / OpenCV / cv_bridge, so it cannot be built here; its logic is
// covered through the ROS-free instantiation in host/ and tests/, and the file
// itself is parsed and type-checked against declaration-only stubs by
// tests/test_abi_cpu.py::test_ros_adaptor_parses -- a syntax check, nothing more).
//
// Same node name, topics, queue sizes, latching and private parameters as the
// reference (src/disparity_to_point_cloud_node.cpp:46-52,
// include/disparity_to_point_cloud/disparity_to_point_cloud.hpp:75-106), so
// launch/d2pcloud.launch works unchanged.  cv_bridge (cpp:50), cv::medianBlur
// (cpp:55-57) and cv::stereoRectify (hpp:104) stay exactly the reference's
// calls; cpp:60-85 is the C-ABI call inside Disparity2PCloudT::DisparityCb.
#include <cv_bridge/cv_bridge.h>
#include <opencv2/calib3d/calib3d.hpp>
#include <opencv2/imgproc/imgproc.hpp>
#include <ros/ros.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/PointCloud2.h>
#include <stereo_msgs/DisparityImage.h>

#include "../host/disparity_to_point_cloud_amd.hpp"
#include "../host/pinned_allocator.hpp"

struct RosMsgs {
  typedef sensor_msgs::Image Image;
  typedef stereo_msgs::DisparityImage DisparityImage;
  // ROS 1 messages are templated on their container allocator: the cloud's byte vector lives in page-locked
  // memory (small members fall through to malloc), and the kernels store the final bytes straight into it
  typedef d2pc::PinnedAllocator<void> CloudAlloc;
  typedef sensor_msgs::PointField_<CloudAlloc> PointField;
  typedef sensor_msgs::PointCloud2_<CloudAlloc> PointCloud2;
  static d2pc::Mono8 prepare(const Image &msg, int median_ksize) {
    cv_bridge::CvImagePtr disparity = cv_bridge::toCvCopy(msg, "mono8");          // cpp:50
    cv::Mat median_filtered(disparity->image.size(), CV_8U);
    if (median_ksize > 1) cv::medianBlur(disparity->image, median_filtered, median_ksize);  // cpp:55-57 (host median)
    else median_filtered = disparity->image.clone();                                        // median runs on the GPU
    d2pc::Mono8 out;
    out.width = median_filtered.cols;
    out.height = median_filtered.rows;
    out.pix.assign(median_filtered.datastart, median_filtered.dataend);            // continuous: freshly allocated
    return out;
  }
};

int main(int argc, char *argv[]) {
  ros::init(argc, argv, "disparity_to_point_cloud");
  ros::NodeHandle nh("~");

  d2pc::ParamSource params;  // ~fx_ ~fy_ ~cx_ ~cy_ ~base_line_ (hpp:84-88)
  for (const char *name : {"fx_", "fy_", "cx_", "cy_", "base_line_"}) {
    double v;
    if (nh.getParam(name, v)) params.values[name] = v;
  }
  // ~reproject_form (not in the reference): which cv::reprojectImageTo3D arithmetic the published bytes reproduce
  // (d2pc_set_reproject_form).  Default: that of the OpenCV THIS node is built against -- 2.4's loop or 3/4's Matx
  // form, bit for bit --, so that replacing the reference node changes no published byte; 0 = the form within 1 ulp
  // of both.
  int form = CV_MAJOR_VERSION >= 3 ? D2PC_FORM_CV4 : D2PC_FORM_CV24;
  nh.param("reproject_form", form, form);
  params.values["reproject_form"] = form;
  double fx = 714.24, fy = 713.5, cx = 376, cy = 240, b = 0.09;
  params.param("fx_", fx, fx); params.param("fy_", fy, fy); params.param("cx_", cx, cx);
  params.param("cy_", cy, cy); params.param("base_line_", b, b);

  // hpp:90-104: Q_ from OpenCV itself, handed to the GPU path as data
  cv::Mat K = (cv::Mat_<double>(3, 3) << fx, 0, cx, 0, fy, cy, 0, 0, 1);
  cv::Mat dist = cv::Mat::zeros(5, 1, CV_64FC1), R = cv::Mat::eye(3, 3, CV_64FC1);
  cv::Mat t = (cv::Mat_<double>(3, 1) << -b, 0, 0), R1, R2, P1, P2, Q;
  cv::stereoRectify(K, dist, K, dist, cv::Size(752, 480), R, t, R1, R2, P1, P2, Q);
  Q.convertTo(Q, CV_64F);

  ros::Publisher pub = nh.advertise<RosMsgs::PointCloud2>("/point_cloud", 1, true);  // hpp:80-81 (latched)
  int device = 0;
  nh.param("device_id", device, 0);  // rank-local GPU when several nodes share a host
  d2pc::Disparity2PCloudT<RosMsgs> node(
      params, [&](const RosMsgs::PointCloud2 &pc) { pub.publish(pc); }, device, Q.ptr<double>());
  ros::Subscriber sub = nh.subscribe<sensor_msgs::Image>(
      "/disparity", 1, [&](const sensor_msgs::ImageConstPtr &msg) { node.DisparityCb(msg); });  // hpp:77-78
  // hpp:65 TODO ("get calibration from the camera"): a stereo_msgs/DisparityImage on an extra topic carries f, T
  // and min_disparity with the 32FC1 disparities; an extra input topic that stays silent unless something
  // publishes on it, so the reference's launch file is unaffected
  ros::Subscriber sub_di = nh.subscribe<stereo_msgs::DisparityImage>(
      "/disparity_image", 1, [&](const stereo_msgs::DisparityImageConstPtr &msg) { node.DisparityImageCb(msg); });
  (void)sub;
  (void)sub_di;
  ros::spin();  // single-threaded, as the reference (node.cpp:50)
  return 0;
}
