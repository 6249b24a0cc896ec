# Builds the three native pieces in-tree:
#   raytracer_amd/lib/librt_hip.so   HIP kernels + C ABI (include/rt_hip.h), gfx950
#   raytracer_amd/lib/librt_host.so  C++ host above the ABI (include/rt_host.h)
#   oracle/_build/librt_oracle.so    CPU restatement (test infrastructure only)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
CC ?= gcc
LIBDIR := raytracer_amd/lib
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function
HOSTSRC := raytracer_amd/csrc/host/scene.cpp raytracer_amd/csrc/host/bvh.cpp raytracer_amd/csrc/host/integrator.cpp

all: hip host oracle

hip: $(LIBDIR)/librt_hip.so
host: $(LIBDIR)/librt_host.so
oracle: oracle/_build/librt_oracle.so

$(LIBDIR)/librt_hip.so: raytracer_amd/csrc/rt_hip.hip raytracer_amd/csrc/rt_device.h raytracer_amd/csrc/rt_build_device.h raytracer_amd/csrc/rt_build_topdown.h raytracer_amd/csrc/rt_comm.h include/rt_hip.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -o $@ raytracer_amd/csrc/rt_hip.hip

$(LIBDIR)/librt_host.so: $(HOSTSRC) raytracer_amd/csrc/host/scene.hpp raytracer_amd/csrc/host/json.hpp include/rt_host.h include/rt_hip.h $(LIBDIR)/librt_hip.so
	@mkdir -p $(LIBDIR)
	$(CXX) -O2 -std=c++17 -fPIC -shared -pthread -ffp-contract=off -Wall -o $@ $(HOSTSRC) -L$(LIBDIR) -lrt_hip -lz -Wl,-rpath,'$$ORIGIN'

# portable on purpose (no -march): this file travels to machines with other CPUs.  The benchmark's CPU-baseline leg
# compiles the same source with -march=native on the machine it measures (oracle/oracle.py: native_build).
oracle/_build/librt_oracle.so: oracle/rt_oracle.c include/rt_hip.h
	@mkdir -p oracle/_build
	$(CC) -O3 -std=gnu11 -fPIC -shared -ffp-contract=off -fno-fast-math -Wall -o $@ oracle/rt_oracle.c -lm -lpthread

clean:
	rm -f $(LIBDIR)/*.so oracle/_build/*.so

.PHONY: all hip host oracle clean
