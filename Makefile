# Builds the shipped library and the CPU checkers.
#
#   make            -> ray-tracer_amd/csrc/librt_amd.so  (HIP, gfx950) + oracle
#   make lib        -> only the shipped library
#   make oracle     -> only the checkers under oracle/ (test infrastructure)
#
# -ffp-contract=off is part of the contract of this library, not a tuning
# choice: results must equal the reference's unfused IEEE arithmetic.

HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := ray-tracer_amd/csrc
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC \
            -fno-fast-math -Wall -Wno-unused-function
LIB      := $(CSRC)/librt_amd.so
SRCS     := $(CSRC)/rt_kernels.hip $(CSRC)/rt_api.hip $(CSRC)/rt_bvh.cpp $(CSRC)/rt_tribvh.cpp
HDRS     := $(CSRC)/rt_device.h $(CSRC)/rt_kernels.h $(CSRC)/rt_bvh.h $(CSRC)/rt_tribvh.h $(CSRC)/rt_walk.h $(CSRC)/rt_trace.h include/rt_amd.h

all: lib oracle

SHIM     := ray-tracer_amd/host/librt_shim.so

HOSTLIB  := ray-tracer_amd/host/librt_host.so
DRIVER   := ray-tracer_amd/host/raytracer_amd

lib: $(LIB) $(SHIM) $(HOSTLIB) $(DRIVER)

# scene-file reader, grid generator, PPM writer (no GPU code)
$(HOSTLIB): ray-tracer_amd/host/scene_io.cpp include/rt_host.h
	g++ -std=c++11 -O2 -ffp-contract=off -fPIC -shared -o $@ ray-tracer_amd/host/scene_io.cpp

# headless driver with the reference's command line
$(DRIVER): ray-tracer_amd/host/rt_main.cpp $(HOSTLIB) $(LIB) include/rt_amd.h include/rt_host.h
	g++ -std=c++11 -O2 -o $@ ray-tracer_amd/host/rt_main.cpp -Lray-tracer_amd/host -lrt_host \
	    -L$(CSRC) -lrt_amd -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,'$$ORIGIN/../csrc'

# The reference's render()/check_intersection() signatures on top of $(LIB)
$(SHIM): ray-tracer_amd/host/render_shim.cpp include/rt_amd.h $(LIB)
	g++ -std=c++11 -O2 -fPIC -shared -o $@ ray-tracer_amd/host/render_shim.cpp \
	    -L$(CSRC) -lrt_amd -Wl,-rpath,'$$ORIGIN/../csrc'

$(LIB): $(SRCS) $(HDRS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(SRCS)

oracle:
	$(MAKE) -C oracle

# diagnostic build with cycle stamps inside the walk kernel (tools/walk_stamps.py);
# never loaded by the product path
stamps: $(SRCS) $(HDRS)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -DRT_STAMPS -shared -o build/librt_amd_stamps.so $(SRCS)

isa: $(SRCS) $(HDRS)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o build/rt_kernels.s $(CSRC)/rt_kernels.hip \
	    -Rpass-analysis=kernel-resource-usage

clean:
	rm -f $(LIB) $(SHIM) $(HOSTLIB) $(DRIVER)
	rm -rf build
	$(MAKE) -C oracle clean

.PHONY: all lib oracle isa stamps clean
