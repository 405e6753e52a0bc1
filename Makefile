# Builds the product library (HIP, gfx950) and the CPU oracle (test infrastructure).
#   make            -> volren_amd/libvolren_amd.so  oracle/liboracle.so  volren_amd/volren
# The HIP sources are cross-compiled; no GPU is needed to build.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := volren_amd/csrc
# -ffp-contract=off: the renderer's fp32 arithmetic is specified operation by operation (vr_math.h)
CXXFLAGS := -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-result -Iinclude
HIPFLAGS := --offload-arch=$(ARCH) $(CXXFLAGS)
OBJDIR   := build
SRCS_CPP := grids.cpp imageio.cpp environment.cpp transferfunc.cpp renderer.cpp sharded.cpp capi.cpp
PT_VARIANTS := 0 1 2 3 4
OBJS     := $(OBJDIR)/vr_kernels.o $(PT_VARIANTS:%=$(OBJDIR)/vr_pathtrace_%.o) $(PT_VARIANTS:%=$(OBJDIR)/vr_ptfast_%.o) $(SRCS_CPP:%.cpp=$(OBJDIR)/%.o)
# tolerance-mode kernels (opt-in, vr_math.h VR_FAST_MATH): hardware transcendentals, reciprocal division, contraction allowed
FASTFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -Wno-unused-result -Iinclude -DVR_FAST_MATH=1
# path-tracing kernels: no SLP vectorisation.  On gfx950 a packed fp32 instruction (v_pk_mul/add/fma_f32) occupies the SIMD for
# 4.2 cycles against 1.8 for the plain one (profiles/r2_valu_issue_rate.txt) and needs extra moves to pair its operands: the
# scalar form of the same arithmetic is 4 % (c2) / 2 % (c3) / 1 % (c4) faster (profiles/r2p_compiler_flags.txt); results identical.
PTFLAGS  := -fno-slp-vectorize
HDRS     := $(wildcard $(CSRC)/*.h) include/volren_amd.h

all: volren_amd/libvolren_amd.so volren_amd/volren oracle

$(OBJDIR)/vr_kernels.o: $(CSRC)/vr_kernels.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# the path-tracing kernel, one compilation per variant (vr_pathtrace.hip); resource usage goes to build/*.resources.txt
$(OBJDIR)/vr_pathtrace_%.o: $(CSRC)/vr_pathtrace.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(PTFLAGS) -DVR_PT_VARIANT=$* -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> $(OBJDIR)/vr_pathtrace_$*.resources.txt || (cat $(OBJDIR)/vr_pathtrace_$*.resources.txt; false)

# (object names with disjoint patterns: vr_pathtrace_% must not also match the tolerance-mode objects)
$(OBJDIR)/vr_ptfast_%.o: $(CSRC)/vr_pathtrace.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(FASTFLAGS) $(PTFLAGS) -DVR_PT_VARIANT=$* -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> $(OBJDIR)/vr_ptfast_$*.resources.txt || (cat $(OBJDIR)/vr_ptfast_$*.resources.txt; false)

$(OBJDIR)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

volren_amd/libvolren_amd.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(OBJS) -lz -ldl

volren_amd/volren: $(CSRC)/main.cpp volren_amd/libvolren_amd.so $(HDRS)
	$(HIPCC) $(HIPFLAGS) -x hip $(CSRC)/main.cpp -o $@ -Lvolren_amd -lvolren_amd -Wl,-rpath,'$$ORIGIN:$$ORIGIN/../lib' -lz

# make install PREFIX=<p>: the C ABI header, the C++ drop-in header with the class headers it includes, the library and the CLI --
#   <p>/include/volren_amd.h  <p>/include/volren_amd.hpp  <p>/include/volren_amd/*.h  <p>/lib/libvolren_amd.so  <p>/bin/volren
# A caller written against the reference's src/renderer.h builds with: hipcc -I<p>/include caller.cpp -L<p>/lib -lvolren_amd
PREFIX ?= /usr/local
# what volren_amd.hpp pulls in (host-side class headers; the kernel headers vr_trace.h / vr_pathtrace.h / vr_math.h stay private)
INSTALL_HDRS := renderer.h environment.h transferfunc.h grids.h sharded.h devmem.h hostmath.h vr_math.h vr_device.h vr_scene.h
install: all
	install -d $(DESTDIR)$(PREFIX)/include/volren_amd $(DESTDIR)$(PREFIX)/lib $(DESTDIR)$(PREFIX)/bin
	install -m 644 include/volren_amd.h include/volren_amd.hpp $(DESTDIR)$(PREFIX)/include/
	install -m 644 $(INSTALL_HDRS:%=$(CSRC)/%) $(DESTDIR)$(PREFIX)/include/volren_amd/
	install -m 755 volren_amd/libvolren_amd.so $(DESTDIR)$(PREFIX)/lib/
	install -m 755 volren_amd/volren $(DESTDIR)$(PREFIX)/bin/

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(OBJDIR) volren_amd/libvolren_amd.so volren_amd/volren
	$(MAKE) -C oracle clean

.PHONY: all oracle clean install
