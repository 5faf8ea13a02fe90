# Builds liboctree_hip.so (gfx950 only).  No cmake: plain hipcc.  (The oracle is NumPy: nothing to build.)
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
# -ffp-contract=off: the parity contract forbids fusing the reference's separate mul/add
HIPFLAGS = -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -ffp-contract=off -Wall -Wno-unused-function -Iinclude
SRCS = $(wildcard octreelib_amd/csrc/*.hip)
HDRS = $(wildcard octreelib_amd/csrc/*.h) include/octreelib_hip.h
OBJS = $(patsubst octreelib_amd/csrc/%.hip,build/%.o,$(SRCS))
LIB  = octreelib_amd/lib/liboctree_hip.so
# test-only stand-in for RCCL (R > 1 rank processes on ONE GPU): tests/test_gpu_route_multirank.py
STUB = tests/rccl_stub/librccl_stub.so

all: $(LIB) $(STUB)

build/%.o: octreelib_amd/csrc/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p octreelib_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@ -ldl

$(STUB): tests/rccl_stub/rccl_stub.cpp
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared $< -o $@ -lrt -lpthread

stub: $(STUB)

clean:
	rm -rf build $(LIB) $(STUB)

.PHONY: all clean stub
