# Top-level convenience targets (the reference builds each directory with a small Makefile too:
# beamformer_coefficient_generator/Makefile).  gfx950 only; hipcc cross-compiles without a GPU.
#
#   make            the C-ABI library, the CPU oracle (test infrastructure) and the host examples
#   make test-cpu   the CPU test-suite (oracle, exhaustive numerics sweeps, host ABI, sharding over gloo)
#   make test-gpu   parity through the C-ABI on an MI355X
#   make bench      python bench.py (one JSON line)
HIPCC   ?= /opt/rocm/bin/hipcc
PYTHON  ?= python
LIB     := dc_sand_amd/csrc/libdcs_beamformer.so
SRCS    := dc_sand_amd/csrc/bf_kernels.hip dc_sand_amd/csrc/bf_beamform_mfma.hip dc_sand_amd/csrc/bf_capi.hip
HDRS    := dc_sand_amd/csrc/bf_kernels.h dc_sand_amd/csrc/bf_math.h dc_sand_amd/csrc/bf_device.h include/dcs_beamformer.h
# -ffp-contract=off is part of the numerical contract; keep in step with dc_sand_amd/build.py
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -fvisibility=hidden \
            -Wall -Wextra -Wno-unused-parameter

all: $(LIB) probes oracle hosts

$(LIB): $(SRCS) $(HDRS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(SRCS)

oracle:
	$(MAKE) -C oracle

# measurement apparatus (include/dcs_probes.h): the probe kernels + a -DDCS_PROBES build of the product sources
probes: probes/libdcs_probes.so
probes/libdcs_probes.so: probes/bf_probes.hip $(SRCS) $(HDRS) include/dcs_probes.h
	$(HIPCC) $(HIPFLAGS) -DDCS_PROBES -shared -o $@ probes/bf_probes.hip $(SRCS)

hosts: $(LIB) oracle
	$(MAKE) -C tests/numerics
	$(MAKE) -C tests/cpp

test-cpu: all
	$(PYTHON) -m pytest tests -x -q -m "not gpu"

test-gpu: all
	$(PYTHON) -m pytest tests -x -q -m gpu

bench: $(LIB)
	$(PYTHON) bench.py

clean:
	rm -f $(LIB) probes/libdcs_probes.so
	$(MAKE) -C oracle clean
	$(MAKE) -C tests/cpp clean
	rm -f tests/numerics/libnumerics_lab.so

.PHONY: all oracle probes hosts test-cpu test-gpu bench clean
