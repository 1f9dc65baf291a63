# Builds the product library (HIP, gfx950 only) and the CPU oracle (tests only).
#   make            -> vettore_amd/lib/libvettore_hip.so + oracle/libvt_oracle.so
# hipcc cross-compiles for gfx950 without a GPU present.
ROCM    ?= /opt/rocm
HIPCC   ?= $(ROCM)/bin/hipcc
ARCH    ?= gfx950
CSRC    := vettore_amd/csrc
LIBDIR  := vettore_amd/lib
# -ffp-contract=off: the reference never fuses a*b+c; the kernels must not either.
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off \
            -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function

# make EXPERIMENTS=1: K2's timing switches (VT_BATCH_DEBUG, tools/batch_debug.sh) are compiled in
ifdef EXPERIMENTS
HIPFLAGS += -DVT_BATCH_TIMING_EXPERIMENTS
endif

DEVSRC  := vt_kernels vt_batch vt_batch_bf16 vt_scan_dot vt_scan_l2 vt_scan_misc vt_scan_general vt_scan_multi
DEVOBJ  := $(addprefix $(LIBDIR)/,$(addsuffix .o,$(DEVSRC)))
DEVHDR  := $(CSRC)/vt_device.h $(CSRC)/vt_common.cuh $(CSRC)/vt_scan.cuh

all: $(LIBDIR)/libvettore_hip.so oracle

$(LIBDIR)/%.o: $(CSRC)/%.hip $(DEVHDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

HOSTHDR := $(wildcard $(CSRC)/host/*.h)
$(LIBDIR)/vt_index.o: $(CSRC)/vt_index.cpp $(HOSTHDR) $(CSRC)/vt_device.h include/vettore_flat.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIBDIR)/libvettore_hip.so: $(DEVOBJ) $(LIBDIR)/vt_index.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lpthread -ldl

oracle:
	$(MAKE) -C oracle -s

# stand-alone hardware probes quoted in DESIGN.md (not part of the library)
PROBES := tools/hbm_peak tools/launch_probe tools/mfma_peak tools/mfma_agpr tools/vmm_probe
probes: $(PROBES)
tools/%: tools/%.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 $< -o $@

clean:
	rm -rf $(LIBDIR) oracle/libvt_oracle.so $(PROBES)

.PHONY: all oracle clean probes
