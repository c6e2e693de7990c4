# Builds the product (libnaf_gpu.so + ennaf/unnaf CLIs) for gfx950 and the test-side oracle.
HIPCC   ?= hipcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value
CSRC     = naf_amd/csrc
OBJS     = $(CSRC)/naf_gpu.o $(CSRC)/scan.o $(CSRC)/zstd_dec.o $(CSRC)/emit.o $(CSRC)/zstd_enc.o $(CSRC)/enc.o $(CSRC)/io.o
HDRS     = $(wildcard $(CSRC)/*.h) include/naf_gpu.h

all: naf_amd/libnaf_gpu.so hosts oracle emul tools/bw_calibrate tools/lds_probe tools/io_probe tests/c/test_gather

# known-byte-count kernels used to calibrate the PMC counters (tools/profile_bench.sh)
tools/bw_calibrate: tools/bw_calibrate.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -w -o $@ $<
# what an LDS operation of a wavefront costs, by kind
tools/lds_probe: tools/lds_probe.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -w -o $@ $<

# what the box can do file <-> HBM (profiles/r03_io_probe.txt)
tools/io_probe: tools/io_probe.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -w -o $@ $< -lpthread

naf_amd/libnaf_gpu.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(OBJS)

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

hosts: naf_amd/bin/ennaf naf_amd/bin/unnaf
naf_amd/bin/%: naf_amd/host/%.c naf_amd/host/host_common.h include/naf_gpu.h naf_amd/libnaf_gpu.so
	@mkdir -p naf_amd/bin
	gcc -O2 -std=gnu99 -Wall -o $@ $< -Lnaf_amd -lnaf_gpu -Wl,-rpath,'$$ORIGIN/..' -Wl,-rpath,/opt/rocm/lib

# a C caller of the decode path's collective (tests/test_gpu_shard.py runs it)
tests/c/test_gather: tests/c/test_gather.c include/naf_gpu.h naf_amd/libnaf_gpu.so
	gcc -O2 -std=gnu99 -Wall -o $@ $< -Lnaf_amd -lnaf_gpu -Wl,-rpath,'$$ORIGIN/../../naf_amd' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -s -C oracle all

emul: tests/emul/libzstd_emul.so
tests/emul/libzstd_emul.so: tests/emul/zstd_emul.cpp tests/emul/zstd_enc_emul.cpp $(CSRC)/zstd_dec_core.h $(CSRC)/zstd_enc_core.h $(CSRC)/enc_swar.h $(CSRC)/common.h
	g++ -O2 -std=c++17 -fPIC -shared -o $@ tests/emul/zstd_emul.cpp tests/emul/zstd_enc_emul.cpp

clean:
	rm -rf $(CSRC)/*.o naf_amd/libnaf_gpu.so tests/emul/*.so naf_amd/bin
.PHONY: all oracle emul hosts clean
