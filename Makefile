# Entry points keep the reference's names (Makefile:1-4 there: `all` builds the host objects
# and the rpv2 binary); everything is built for gfx950 only.
#
#   make / make all   libwrp.so (C ABI + HIP kernels), host objects, rpv2 binary, oracle
#   make lib          weather-radar-processing_amd/lib/libwrp.so only
#   make oracle       oracle/liboracle.so (+ oracle/_ref when /root/reference exists)
#   make process      the reference's commented alternative entry point (Makefile:3 there: `process`, the UDP program of
#                     gpu_1fp_streamcasc.cu / read_single.cc): the same binary as rpv2, under that name -- run it as
#                     `process 4 --in udp:19001 --out udp:19002,19003` (its defaults)
#   make clean

HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
PKG      := weather-radar-processing_amd
CSRC     := $(PKG)/csrc
HOST     := $(PKG)/host
LIBDIR   := $(PKG)/lib
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-function -ffp-contract=off -fno-slp-vectorize
CXXFLAGS ?= -O2 -std=c++17 -fPIC -Wall

all: lib host oracle process

lib: $(LIBDIR)/libwrp.so

$(LIBDIR)/libwrp.so: $(CSRC)/wrp_engine.hip $(CSRC)/wrp_kernels.h $(CSRC)/wrp_generic.h $(CSRC)/wrp_fused.h $(CSRC)/wrp_shape_b.h $(CSRC)/wrp_fused_b.h $(CSRC)/fft_radix.h include/wrp.h
	mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/wrp_engine.hip

host:
	@if [ -f $(HOST)/Makefile ]; then $(MAKE) -C $(HOST); fi

process: host
	ln -sf rpv2 $(HOST)/process

oracle:
	$(MAKE) -C oracle
	@if [ -d /root/reference ]; then $(MAKE) -C oracle ref; fi

clean:
	rm -rf $(LIBDIR)
	$(MAKE) -C oracle clean
	@if [ -f $(HOST)/Makefile ]; then $(MAKE) -C $(HOST) clean; fi

.PHONY: all lib host oracle process clean
