run() { python tools/ab.py --libs base,ctlprio --modes compact --algos 2 --pxts 8 --opbpc 0 --rounds 9 --iters 20 "$@" 2>&1 | grep -v amdgpu.ids | sed 's/ b=40 pxt= 8 bpc=128 novec=0 algo=2 oalign=16 ooff=0 form=0//'; }
echo "== 16 x 4K, 30 % holes"; run --holes 0.3 --idx 0
echo "== 16 x 4K, 30 % holes + indices"; run --holes 0.3 --idx 1
echo "== 16 x 4K, all valid"; run --holes 0 --idx 0
echo "== 32 x 1080p, 30 % holes + indices"; run --holes 0.3 --idx 1 --frames 32 --w 1920 --h 1080
