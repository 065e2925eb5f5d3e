for args in "--frames 1 --holes 0 --idx 1" "--frames 1 --holes 0.3 --idx 0" "--frames 1 --holes 0.6 --idx 0" "--frames 1 --holes 0.9 --idx 0" "--frames 1 --holes 0.3 --blocky 1 --idx 0"; do
  echo "== $args"
  python tools/ab.py --modes compact --algos 1,3 --pxts 8 --rounds 9 --iters 20 --tunes "resident_pxt=32" $args 2>&1 | grep -v amdgpu.ids
done
