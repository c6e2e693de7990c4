# Builds the product (libnaf_gpu.so + ennaf/unnaf CLIs) for gfx950 and the test-side oracle.
HIPCC   ?= hipcc
ARCH    ?= gfx950
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value
CSRC     = naf_amd/csrc
OBJS     = $(CSRC)/naf_gpu.o $(CSRC)/scan.o $(CSRC)/zstd_dec.o $(CSRC)/emit.o $(CSRC)/zstd_enc.o $(CSRC)/enc.o
HDRS     = $(wildcard $(CSRC)/*.h) include/naf_gpu.h

all: naf_amd/libnaf_gpu.so oracle

naf_amd/libnaf_gpu.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(OBJS)

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

oracle:
	$(MAKE) -s -C oracle all

emul: tests/emul/libzstd_emul.so
tests/emul/libzstd_emul.so: tests/emul/zstd_emul.cpp $(CSRC)/zstd_dec_core.h $(CSRC)/common.h
	g++ -O2 -std=c++17 -fPIC -shared -o $@ tests/emul/zstd_emul.cpp

clean:
	rm -f $(CSRC)/*.o naf_amd/libnaf_gpu.so tests/emul/*.so
.PHONY: all oracle emul clean
