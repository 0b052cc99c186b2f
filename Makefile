# Builds libstarkhip.so (HIP kernels for gfx950 + host C++) and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := starky_bls12_381_amd/csrc
OUT := starky_bls12_381_amd/libstarkhip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude
SRCS := $(wildcard $(CSRC)/*.hip) $(wildcard $(CSRC)/*.cpp)
OBJS := $(patsubst $(CSRC)/%,build/%.o,$(SRCS))
HDRS := $(wildcard $(CSRC)/*.h) include/starkhip.h

all: $(OUT) oracle demo

build/%.hip.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
build/%.cpp.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p build
	$(HIPCC) -O2 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude -c $< -o $@

$(OUT): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle

# C++ drivers above the C ABI (include/starkhip_driver.hpp): the six proofs of one signature check
demo: build/signature_demo
build/signature_demo: tools/signature_demo.cpp include/starkhip_driver.hpp $(OUT)
	@mkdir -p build
	g++ -O2 -std=c++17 -Iinclude $< -o $@ -Lstarky_bls12_381_amd -lstarkhip -Wl,-rpath,'$$ORIGIN/../starky_bls12_381_amd'

clean:
	rm -rf build $(OUT)
	$(MAKE) -C oracle clean
.PHONY: all oracle clean demo
