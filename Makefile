# Builds libstarkhip.so (HIP kernels for gfx950 + host C++) and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := starky_bls12_381_amd/csrc
OUT := starky_bls12_381_amd/libstarkhip.so
ROCM_PATH ?= /opt/rocm
# make ROCTX=1: rocTX phase ranges inside prove() (links the profiler SDK's roctx library); make DEBUG_KNOBS=1: the profiling
# modes of the quotient kernel that switch arithmetic off ("quotient_debug" option) -- neither is in the default library
EXTRA_DEFS := $(if $(ROCTX),-DSTARKHIP_ROCTX) $(if $(DEBUG_KNOBS),-DSTARKHIP_DEBUG)
EXTRA_LIBS := $(if $(ROCTX),-L$(ROCM_PATH)/lib -lrocprofiler-sdk-roctx)
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude $(EXTRA_DEFS)
SRCS := $(wildcard $(CSRC)/*.hip) $(wildcard $(CSRC)/*.cpp)
OBJS := $(patsubst $(CSRC)/%,build/%.o,$(SRCS))
HDRS := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.inc) include/starkhip.h

all: $(OUT) oracle demo

# FLAGS_<file>: compiler flags of ONE translation unit (e.g. FLAGS_kernels_lde), in the default and in the variant build
build/%.hip.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(FLAGS_$*) -c $< -o $@
build/%.cpp.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p build
	$(HIPCC) -O2 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude $(EXTRA_DEFS) -c $< -o $@

$(OUT): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -lpthread $(EXTRA_LIBS)

oracle:
	$(MAKE) -C oracle

# C++ drivers above the C ABI (include/starkhip_driver.hpp): the six proofs of one signature check
demo: build/signature_demo
build/signature_demo: tools/signature_demo.cpp include/starkhip_driver.hpp $(OUT)
	@mkdir -p build
	g++ -O2 -std=c++17 -Iinclude $< -o $@ -Lstarky_bls12_381_amd -lstarkhip -Wl,-rpath,'$$ORIGIN/../starky_bls12_381_amd'

# Sanitizer build of the HOST code (AddressSanitizer + UndefinedBehaviorSanitizer; GPU sanitizers are not available on this
# pool): csrc/*.cpp + stand-ins for the device entry points -> build/asan/libstarkhip_host_asan.so, and the oracle likewise.
# `make asan-test` runs the CPU tests that exercise natives, trace generators, AIR builders, verifier and plan builder on it.
ASAN_FLAGS := -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -Iinclude -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
ASAN_SRCS := $(wildcard $(CSRC)/*.cpp) $(CSRC)/host_only_stubs.cc
asan: build/asan/libstarkhip_host_asan.so build/asan/liboracle_asan.so
build/asan/libstarkhip_host_asan.so: $(ASAN_SRCS) $(HDRS)
	@mkdir -p build/asan
	g++ $(ASAN_FLAGS) -shared -o $@ $(ASAN_SRCS)
build/asan/liboracle_asan.so: oracle/stark_oracle.c oracle/oracle_field.h
	@mkdir -p build/asan
	gcc -O1 -g -fopenmp -fPIC -fvisibility=hidden -std=gnu11 -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $@ oracle/stark_oracle.c
asan-test: asan
	LD_PRELOAD=$$(gcc -print-file-name=libasan.so):$$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
	STARKHIP_LIBRARY=$(CURDIR)/build/asan/libstarkhip_host_asan.so STARKHIP_ORACLE_LIBRARY=$(CURDIR)/build/asan/liboracle_asan.so \
	python -m pytest tests/test_native_cpu.py tests/test_quotient_plan_cpu.py tests/test_toy_air_cpu.py tests/test_trace_log_cpu.py tests/test_ecc_aggregate_cpu.py -x -q -m "not gpu" -p no:cacheprovider

# ThreadSanitizer on the threaded trace recording (trace_tasks.cpp): three traces recorded on 6 threads
tsan-test: tests/tsan_trace_main.cpp $(ASAN_SRCS) $(HDRS)
	@mkdir -p build/tsan
	g++ -O1 -g -std=c++17 -fsanitize=thread -fno-omit-frame-pointer -Iinclude -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o build/tsan/tsan_trace tests/tsan_trace_main.cpp $(ASAN_SRCS) -lpthread
	TSAN_OPTIONS=halt_on_error=1 build/tsan/tsan_trace
	g++ -O1 -g -std=c++17 -fsanitize=thread -fno-omit-frame-pointer -Iinclude -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o build/tsan/tsan_pool tests/tsan_pool_main.cpp $(ASAN_SRCS) -lpthread
	TSAN_OPTIONS=halt_on_error=1 build/tsan/tsan_pool
# the same pool harness under AddressSanitizer + UndefinedBehaviorSanitizer
asan-pool-test: tests/tsan_pool_main.cpp $(ASAN_SRCS) $(HDRS)
	@mkdir -p build/asan
	g++ $(filter-out -shared -fPIC,$(ASAN_FLAGS)) -o build/asan/asan_pool tests/tsan_pool_main.cpp $(ASAN_SRCS) -lpthread
	ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 build/asan/asan_pool

# Experiment builds of the device code for A/B runs on one box: `make variant NAME=x DEFS="-D..."` -> build/x/libstarkhip_x.so, selected
# at run time with STARKHIP_LIBRARY=build/x/libstarkhip_x.so (e.g. NAME=nomfma DEFS=-DSTARKHIP_LANE_NO_MFMA: the lane-form leaf hash with every
# round as multiply-add chains; NAME=noprio DEFS=-DSTARKHIP_NO_PRIO: no raised issue priority for the kernels beside the lane-form hash)
NAME ?= variant
VAR_OBJS := $(patsubst $(CSRC)/%.hip,build/$(NAME)/%.hip.o,$(wildcard $(CSRC)/*.hip))
build/$(NAME)/%.hip.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p build/$(NAME)
	$(HIPCC) $(HIPFLAGS) $(DEFS) $(FLAGS_$*) -c $< -o $@
variant: build/$(NAME)/libstarkhip_$(NAME).so
build/$(NAME)/libstarkhip_$(NAME).so: $(VAR_OBJS) $(filter %.cpp.o,$(OBJS))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lpthread $(EXTRA_LIBS)

clean:
	rm -rf build $(OUT)
	$(MAKE) -C oracle clean
.PHONY: all oracle clean demo asan asan-test tsan-test asan-pool-test variant
