# time per step of one bundle (slope of the apply time over the row length), product build and ablated builds
for v in default "$@"; do
  for nx in 128 512 2048; do
    if [ $v = default ]; then python3 tools/micro/mic_ablate.py default $nx,8,8 50 2>&1 | tail -1; else python3 tools/micro/mic_ablate.py tools/micro/_abl/libmanta_abl$v.so $nx,8,8 50 2>&1 | tail -1; fi
  done
done
