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

# `make experiments`: the same library with the timing experiments of K2 / K2b / K2s / K1m compiled in and their
# settings named (batch_debug, mq_dbg, shadow_stages, trace_batch: csrc/vt_env.h) -> vettore_amd/lib/experiments/
# libvettore_hip.so.  Wrong results on purpose (barriers removed, stages skipped): for the cost breakdowns in
# DESIGN_APPENDIX only, never loaded by a test.  The product build carries neither the switches nor the code behind them.
# Kernels that must not spill are checked after they are compiled (tools/check_scratch.py says why); the experiments' extra
# registers do spill, and nobody ships that build.
CHECK_SCRATCH := python3 tools/check_scratch.py
ifdef EXPERIMENTS
HIPFLAGS += -DVT_EXPERIMENTS -DVT_BATCH_TIMING_EXPERIMENTS -DVT_MULTI_TIMING_EXPERIMENTS
CHECK_SCRATCH := @true
endif

DEVSRC  := vt_kernels vt_batch vt_batch_bf16 vt_batch_shadow vt_scan_dot vt_scan_l2 vt_scan_l1 vt_scan_misc vt_scan_general vt_scan_gather vt_scan_multi vt_prefix_multi
DEVOBJ  := $(addprefix $(LIBDIR)/,$(addsuffix .o,$(DEVSRC)))
DEVHDR  := $(CSRC)/vt_device.h $(CSRC)/vt_common.cuh $(CSRC)/vt_scan.cuh $(CSRC)/vt_env.h

all: $(LIBDIR)/libvettore_hip.so $(LIBDIR)/libvettore_hip_hooks.so $(LIBDIR)/libvt_callers.so $(LIBDIR)/libvt_callers_hooks.so oracle

$(LIBDIR)/%.o: $(CSRC)/%.hip $(DEVHDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# K1's instantiation units: no scratch at all (r05: the overflow recovery returns its value by value; every launch of a
# kernel with a scratch segment has it set up)
$(LIBDIR)/vt_scan_%.o: $(CSRC)/vt_scan_%.hip $(DEVHDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> $(LIBDIR)/vt_scan_$*.resources
	$(CHECK_SCRATCH) --no-scratch $(LIBDIR)/vt_scan_$*.resources scan_topk_kernel

# Kernels that must not spill (tools/check_scratch.py says why): the build fails if one does.
$(LIBDIR)/vt_scan_multi.o: $(CSRC)/vt_scan_multi.hip $(DEVHDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> $(LIBDIR)/vt_scan_multi.resources
	$(CHECK_SCRATCH) --no-scratch $(LIBDIR)/vt_scan_multi.resources scan_multi_kernel
$(LIBDIR)/vt_batch_bf16.o: $(CSRC)/vt_batch_bf16.hip $(DEVHDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> $(LIBDIR)/vt_batch_bf16.resources
	$(CHECK_SCRATCH) $(LIBDIR)/vt_batch_bf16.resources bf16_scores_kernel
# (-fno-slp-vectorize: K1p packs two ELEMENTS of a query per instruction by hand; the SLP pass re-packs two QUERIES
# instead, with a register move per operand pair -- 22-26 % more VALU cycles per chunk, measured)
$(LIBDIR)/vt_prefix_multi.o: $(CSRC)/vt_prefix_multi.hip $(DEVHDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> $(LIBDIR)/vt_prefix_multi.resources
	$(CHECK_SCRATCH) --no-scratch $(LIBDIR)/vt_prefix_multi.resources prefix_multi_kernel
$(LIBDIR)/vt_batch_shadow.o: $(CSRC)/vt_batch_shadow.hip $(DEVHDR)
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> $(LIBDIR)/vt_batch_shadow.resources
	$(CHECK_SCRATCH) $(LIBDIR)/vt_batch_shadow.resources shadow_scores_kernel

HOSTHDR := $(wildcard $(CSRC)/host/*.h)
$(LIBDIR)/vt_index.o: $(CSRC)/vt_index.cpp $(HOSTHDR) $(CSRC)/vt_device.h $(CSRC)/vt_env.h include/vettore_flat.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

# (only the C ABI is exported: csrc/exports.map)
EXPORTS := -Wl,--version-script=$(CSRC)/exports.map
$(LIBDIR)/libvettore_hip.so: $(DEVOBJ) $(LIBDIR)/vt_index.o $(CSRC)/exports.map
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(filter %.o,$^) $(EXPORTS) -lpthread -ldl

# The same library with the fault-injection hooks compiled into the host side (VT_TEST_FAIL_AFTER_ID_UPDATE,
# VT_TEST_FOREIGN_ROWS): test infrastructure, loaded only by the two tests that need them
# (VETTORE_HIP_LIB=...); the product library carries no such switch.
$(LIBDIR)/vt_index_hooks.o: $(CSRC)/vt_index.cpp $(HOSTHDR) $(CSRC)/vt_device.h $(CSRC)/vt_env.h include/vettore_flat.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -DVT_TEST_HOOKS -x hip -c $< -o $@

$(LIBDIR)/libvettore_hip_hooks.so: $(DEVOBJ) $(LIBDIR)/vt_index_hooks.o $(CSRC)/exports.map
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(filter %.o,$^) $(EXPORTS) -lpthread -ldl

experiments:
	$(MAKE) EXPERIMENTS=1 LIBDIR=vettore_amd/lib/experiments vettore_amd/lib/experiments/libvettore_hip.so

# bench.py's native caller threads (tools/callers_native.cpp): measurement infrastructure, not product
$(LIBDIR)/libvt_callers.so: tools/callers_native.cpp include/vettore_flat.h $(LIBDIR)/libvettore_hip.so
	g++ -O2 -std=c++17 -fPIC -shared tools/callers_native.cpp -Iinclude -L$(LIBDIR) -lvettore_hip -lpthread -Wl,-rpath,'$$ORIGIN' -o $@

# the same callers bound to the hooks build (tests that force callers to meet: vt_callers_meet with `hold`)
$(LIBDIR)/libvt_callers_hooks.so: tools/callers_native.cpp include/vettore_flat.h $(LIBDIR)/libvettore_hip_hooks.so
	g++ -O2 -std=c++17 -fPIC -shared tools/callers_native.cpp -Iinclude -L$(LIBDIR) -lvettore_hip_hooks -lpthread -Wl,-rpath,'$$ORIGIN' -o $@

oracle:
	$(MAKE) -C oracle -s

# stand-alone hardware probes quoted in DESIGN.md (not part of the library)
PROBES := tools/hbm_peak tools/launch_probe tools/mfma_peak tools/mfma_agpr tools/vmm_probe
probes: $(PROBES)
tools/%: tools/%.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 $< -o $@

clean:
	rm -rf $(LIBDIR) oracle/libvt_oracle.so $(PROBES)

.PHONY: all oracle clean probes experiments
