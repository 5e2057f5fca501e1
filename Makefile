# Builds everything in-tree for gfx950 (MI355X).  No GPU is needed to build.
#   pathtracer_amd/lib/libmi_tracer.so  HIP kernels + C-ABI (include/mi_tracer.h)
#   pathtracer_amd/lib/libpt_host.so    host side: .ray parser, BVH build, Tracer/HipTracer
#   pathtracer_amd/bin/ray              command-line renderer
#   oracle/liboracle.so                 CPU oracle (test infrastructure only)
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
ARCH ?= gfx950
# -ffp-contract=off: the numerics contract — no FMA contraction, so
# device results match the oracle bit for bit.
# -fno-slp-vectorize: left to itself the SLP vectorizer pairs scalar fp32 operations of the traversal and shading code into
# v_pk_mul_f32 / v_pk_add_f32 (same results; IEEE either way), which costs every workload 2-3 % (C3 5.95 -> 5.81 ms, C2 -3.3 %).
HIPFLAGS = --offload-arch=$(ARCH) -O3 -fno-slp-vectorize -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $(EXTRA_HIPFLAGS)
# make WIDE8=1: the 8-wide node format (mi_trav8.h, mi_wide8.h) instead of the quantised 4-wide one
ifeq ($(WIDE8),1)
HIPFLAGS += -DMI_WIDE8
endif
CXXFLAGS = -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -pthread
LIB = pathtracer_amd/lib
BIN = pathtracer_amd/bin
HOST_SRC = pathtracer_amd/host/scene.cpp pathtracer_amd/host/ray_parser.cpp pathtracer_amd/host/tracer.cpp \
           pathtracer_amd/host/image_io.cpp pathtracer_amd/host/host_capi.cpp
HOST_HDR = pathtracer_amd/host/scene.h pathtracer_amd/host/tracer.h pathtracer_amd/host/vec4.h include/mi_tracer.h

all: $(LIB)/libmi_tracer.so $(LIB)/libpt_host.so $(BIN)/ray oracle/liboracle.so

$(LIB)/libmi_tracer.so: pathtracer_amd/csrc/mi_tracer.hip pathtracer_amd/csrc/mi_kernels.h pathtracer_amd/csrc/mi_bvh_build.h pathtracer_amd/csrc/mi_trav8.h pathtracer_amd/csrc/mi_wide8.h include/mi_tracer.h
	@mkdir -p $(LIB)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ pathtracer_amd/csrc/mi_tracer.hip

$(LIB)/libpt_host.so: $(HOST_SRC) $(HOST_HDR) $(LIB)/libmi_tracer.so
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -L$(LIB) -lmi_tracer -lz -Wl,-rpath,'$$ORIGIN'

$(BIN)/ray: pathtracer_amd/host/ray_main.cpp $(LIB)/libpt_host.so
	@mkdir -p $(BIN)
	$(CXX) $(CXXFLAGS) -o $@ pathtracer_amd/host/ray_main.cpp -L$(LIB) -lpt_host -lmi_tracer -Wl,-rpath,'$$ORIGIN/../lib'

oracle/liboracle.so: oracle/oracle.cpp include/mi_tracer.h
	$(MAKE) -C oracle

# The same libraries with the 8-wide node format, beside the default build: MI_TRACER_LIB_DIR=pathtracer_amd/lib8 selects them
LIB8 = pathtracer_amd/lib8
wide8: $(LIB8)/libmi_tracer.so $(LIB8)/libpt_host.so
$(LIB8)/libmi_tracer.so: pathtracer_amd/csrc/mi_tracer.hip pathtracer_amd/csrc/mi_kernels.h pathtracer_amd/csrc/mi_bvh_build.h pathtracer_amd/csrc/mi_trav8.h pathtracer_amd/csrc/mi_wide8.h include/mi_tracer.h
	@mkdir -p $(LIB8)
	$(HIPCC) $(HIPFLAGS) -DMI_WIDE8 -shared -o $@ pathtracer_amd/csrc/mi_tracer.hip
$(LIB8)/libpt_host.so: $(HOST_SRC) $(HOST_HDR) $(LIB8)/libmi_tracer.so
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -L$(LIB8) -lmi_tracer -lz -Wl,-rpath,'$$ORIGIN'

clean:
	rm -rf $(LIB) $(LIB8) $(BIN) oracle/liboracle.so

.PHONY: all clean wide8
