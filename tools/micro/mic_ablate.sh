# timing experiments on the row-streaming MIC sweeps: builds of mic.hip with -DROWS_ABLATE=k next to the product library
# (tools/micro/_abl/, built by hand: see the hipcc lines in DESIGN section 6 / the commit that added this file), run from the repo root
for dims in 256,8,8 256,16,16 256,64,64 256,256,256; do
  python3 tools/micro/mic_ablate.py default $dims 30 2>&1 | tail -1
  for v in "$@"; do python3 tools/micro/mic_ablate.py tools/micro/_abl/libmanta_abl$v.so $dims 30 2>&1 | tail -1; done
done
