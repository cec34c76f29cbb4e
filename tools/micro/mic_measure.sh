# Measurement builds of the row-streaming MIC sweeps (run in the build container from the repo root; the product library must be built):
# applies tools/micro/mic_measure.diff to a scratch copy of mic.hip and links experiment libraries next to the product objects --
#   tools/micro/_abl/libmanta_abl<k>.so   -DROWS_ABLATE=k   (timing only, wrong results: tools/micro/mic_ablate.sh, mic_slope.sh)
#   tools/micro/_abl/libmanta_trace.so    -DROWS_TRACE=1    (time stamps of every hand-off: tools/micro/mic_trace.py)
# plus the stand-alone microbenchmarks valu_chain, handoff_pingpong, lds_dma.  tools/micro/_abl/ is scratch (git-ignored).
set -e
C=mantaflow_amd/csrc
O=tools/micro/_abl
mkdir -p $O
cp $C/mic.hip $O/mic.hip && cp $C/common.h $C/pressure.h $O/ && (cd $O && patch -s -p3 mic.hip < ../mic_measure.diff)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -I$C -Iinclude"
OBJS="$C/build/runtime.o $C/build/pressure.o $C/build/advect.o $C/build/flip.o $C/build/glue.o $C/build/surface.o $C/build/p2g_ordered.o $C/build/noise.o $C/build/turbulence.o"
for k in ${ABLATE:-1 4 5 17 65 113 133}; do
  /opt/rocm/bin/hipcc $FLAGS -DROWS_ABLATE=$k -c $O/mic.hip -o $O/mic_$k.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libmanta_abl$k.so $OBJS $O/mic_$k.o &
done
/opt/rocm/bin/hipcc $FLAGS -DROWS_TRACE=1 -c $O/mic.hip -o $O/mic_trace.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libmanta_trace.so $OBJS $O/mic_trace.o &
for b in valu_chain handoff_pingpong lds_dma; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -w tools/micro/$b.hip -o $O/$b & done
wait
ls $O
