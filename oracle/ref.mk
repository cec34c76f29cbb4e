# oracle/ref.mk -- TEST INFRASTRUCTURE ONLY.
#
# Builds the reference's own C++ hot path (zoharl3/mantaflow, read-only at $(REF)) into
# oracle/_ref/libmanta_ref.so, from the sources where they lie, with g++ directly:
#   1. the reference's `prep` code generator is compiled from $(REF)/source/preprocessor/*.cpp
#      (with -DNOPYTHON=1, the reference's own "no python" packaging, CMakeLists.txt:79,141,316-337);
#   2. prep expands every KERNEL()/PYTHON() source the path touches into oracle/_ref/build/pp/
#      (same command line the reference build uses: `prep generate 0 OPENMP <srcdir>/ <file> <out>`);
#   3. gitinfo.h is produced by the reference's own tools/getGitVersion.py (CMakeLists.txt:577-585);
#   4. the expanded sources + $(REF)/source/nopython/pclass.cpp + $(REF)/source/util/{vectorbase,vector4d,
#      simpleimage}.cpp + $(REF)/dependencies/cnpy/cnpy.cpp (NOPP_SOURCES, CMakeLists.txt:320-325,558-571) and
#      our shim oracle/ref_shim.cpp (our code: a C ABI over the reference's classes) are linked.
# The reference's cmake build is NOT run.  No reference file is copied into the repository: every
# intermediate lives under oracle/_ref/build/ (git-ignored AND gpurun-ignored); only the .so travels.
#
# Flags follow the reference's Release build: -O3 -DNDEBUG, OpenMP, fp32 Real, no -march (so no FMA
# contraction on x86-64).
REF     ?= /root/reference
OUT     := $(dir $(lastword $(MAKEFILE_LIST)))_ref
B       := $(OUT)/build
PP      := $(B)/pp/source
CXX     ?= g++
CXXFLAGS := -O3 -DNDEBUG -DNOPYTHON=1 -DMANTA_MT=1 -DOPENMP=1 -fopenmp -fPIC -std=c++14 -w
INC     := -I$(PP) -I$(PP)/util -I$(PP)/fileio -I$(REF)/source/nopython -I$(REF)/source/util \
           -I$(REF)/source/fileio -I$(REF)/dependencies/cnpy

# sources that need the reference's preprocessor (subset of PP_SOURCES/PP_HEADERS, CMakeLists.txt:176-247)
PP_CPP := general.cpp fluidsolver.cpp conjugategrad.cpp multigrid.cpp grid.cpp grid4d.cpp levelset.cpp \
          fastmarch.cpp shapes.cpp mesh.cpp particle.cpp movingobs.cpp noisefield.cpp kernel.cpp timing.cpp \
          vortexsheet.cpp vortexpart.cpp turbulencepart.cpp edgecollapse.cpp \
          fileio/ioutil.cpp fileio/iogrids.cpp fileio/iomeshes.cpp fileio/ioparticles.cpp fileio/iovdb.cpp \
          fileio/mantaio.cpp \
          plugin/advection.cpp plugin/extforces.cpp plugin/flip.cpp plugin/initplugins.cpp plugin/pressure.cpp \
          plugin/ptsplugins.cpp plugin/waveletturbulence.cpp plugin/apic.cpp
PP_H   := general.h commonkernels.h conjugategrad.h multigrid.h fastmarch.h fluidsolver.h grid.h grid4d.h \
          mesh.h particle.h levelset.h shapes.h noisefield.h vortexsheet.h kernel.h timing.h movingobs.h \
          fileio/mantaio.h edgecollapse.h vortexpart.h turbulencepart.h

GEN_CPP := $(addprefix $(PP)/,$(PP_CPP))
GEN_H   := $(addprefix $(PP)/,$(PP_H))
OBJS    := $(patsubst $(PP)/%.cpp,$(B)/obj/%.o,$(GEN_CPP)) $(B)/obj/nopython_pclass.o $(B)/obj/vectorbase.o $(B)/obj/vector4d.o \
           $(B)/obj/simpleimage.o $(B)/obj/cnpy.o $(B)/obj/ref_shim.o

all: $(OUT)/libmanta_ref.so

$(B)/prep: $(wildcard $(REF)/source/preprocessor/*.cpp)
	@mkdir -p $(B)
	$(CXX) -O2 -w -DNOPYTHON=1 $^ -o $@

$(PP)/%: $(REF)/source/% $(B)/prep
	@mkdir -p $(dir $@)
	$(B)/prep generate 0 OPENMP $(REF)/source/ $* $@ > /dev/null

$(PP)/gitinfo.h:
	@mkdir -p $(PP)
	cd $(REF) && python3 tools/getGitVersion.py $(abspath $@) > /dev/null

$(B)/obj/%.o: $(PP)/%.cpp $(GEN_H) $(PP)/gitinfo.h
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) $(INC) -c $< -o $@

$(B)/obj/nopython_pclass.o: $(REF)/source/nopython/pclass.cpp $(GEN_H)
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) $(INC) -c $< -o $@
$(B)/obj/vectorbase.o $(B)/obj/vector4d.o $(B)/obj/simpleimage.o: $(B)/obj/%.o: $(REF)/source/util/%.cpp $(GEN_H)
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) $(INC) -c $< -o $@
$(B)/obj/cnpy.o: $(REF)/dependencies/cnpy/cnpy.cpp
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) $(INC) -c $< -o $@
$(B)/obj/ref_shim.o: $(dir $(lastword $(MAKEFILE_LIST)))ref_shim.cpp $(GEN_H)
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) $(INC) -c $< -o $@

$(OUT)/libmanta_ref.so: $(OBJS)
	$(CXX) -shared -fopenmp -Wl,-z,defs -o $@ $(OBJS) -lz

clean:
	rm -rf $(OUT)
.PHONY: all clean
.SECONDARY:
