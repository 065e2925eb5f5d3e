run() { python tools/ab.py --libs base --modes compact --algos 2 --pxts 8 --rounds 7 --iters 10 --holes 0.3 --idx 1 "$@" 2>&1 | grep -v amdgpu.ids | sed 's/ b=40 pxt= 8 bpc=128 novec=0 algo=2 oalign=16 ooff=0 form=0//'; }
for F in 8 16 32 64 128; do echo "== 1080p x $F (blocks/CU 3, 4, 6)"; for B in 3 4 6; do run --w 1920 --h 1080 --frames $F --opbpc $B; done; done
for F in 4 8 16 32; do echo "== 4K x $F (blocks/CU 3, 4)"; for B in 3 4; do run --frames $F --opbpc $B; done; done
