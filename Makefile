# Builds the product library (HIP, gfx950 only) and the CPU oracle (tests only).
#   make            -> vettore_amd/lib/libvettore_hip.so + oracle/libvt_oracle.so
# hipcc cross-compiles for gfx950 without a GPU present.
ROCM    ?= /opt/rocm
HIPCC   ?= $(ROCM)/bin/hipcc
ARCH    ?= gfx950
CSRC    := vettore_amd/csrc
LIBDIR  := vettore_amd/lib
# -ffp-contract=off: the reference never fuses a*b+c; the kernels must not either.
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function

all: $(LIBDIR)/libvettore_hip.so oracle

$(LIBDIR)/vt_device.o: $(CSRC)/vt_device.hip $(CSRC)/vt_device.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/vt_index.o: $(CSRC)/vt_index.cpp $(CSRC)/vt_device.h include/vettore_flat.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIBDIR)/libvettore_hip.so: $(LIBDIR)/vt_device.o $(LIBDIR)/vt_index.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lpthread

oracle:
	$(MAKE) -C oracle -s

clean:
	rm -rf $(LIBDIR) oracle/libvt_oracle.so

.PHONY: all oracle clean
