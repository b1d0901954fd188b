# Builds libqdrant_amd.so (HIP, gfx950 only) and the CPU oracle (test infrastructure).
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# (--offload-compress: the gfx950 code objects are zstd-compressed inside the fat binary and unpacked by the HIP runtime when the library is loaded:
#  the same kernels in a third of the bytes)
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden --offload-compress \
            -Wall -Wno-unused-function -Iinclude
CSRC     := qdrant_amd/csrc
SRCS     := $(wildcard $(CSRC)/*.hip)
OBJS     := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS))
DEPS     := $(OBJS:.o=.d)
LIB      := qdrant_amd/libqdrant_amd.so
# synthetic row generators of bench.py / tools / tests: their own library, not part of the product
TESTDATA := qdrant_amd/libqmx_testdata.so

all: $(LIB) $(TESTDATA) oracle

$(TESTDATA): qdrant_amd/testdata/synth.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -shared -o $@ $<

# (-MMD: every object depends on the headers IT includes - a change to hnsw.hpp rebuilds the walk's translation units, not the scans)
build/%.o: $(CSRC)/%.hip
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -MMD -MP -c $< -o $@
-include $(DEPS)

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIB) $(TESTDATA)
	$(MAKE) -C oracle clean
.PHONY: all oracle clean
