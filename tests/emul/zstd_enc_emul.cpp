// zstd_enc_emul.cpp -- DEVELOPMENT/TEST HARNESS (host build of the encoder's per-lane logic).
#include "../../naf_amd/csrc/zstd_enc_core.h"
#include <vector>

extern "C" long long emul_zstd_compress(const u8 *src, size_t n, u32 block, u8 *dst, size_t cap)
{
    u8 *p = dst;
    *p++ = 0x28; *p++ = 0xB5; *p++ = 0x2F; *p++ = 0xFD; *p++ = 0x00; *p++ = 0x58;   // FHD 0, windowLog 21
    size_t nblk = n ? (n + block - 1) / block : 1;
    for (size_t b = 0; b < nblk; b++) {
        const u8 *s = src + b * block; u32 bn = (u32)(n - b * block < block ? n - b * block : block);
        std::vector<u32> hist(1024, 0);
        u32 per = (bn + 3) / 4;
        for (u32 i = 0; i < bn; i++) hist[256 * (i / (per ? per : 1) > 3 ? 3 : i / (per ? per : 1)) + s[i]]++;
        ZEncPlan pl; u8 len[256], tree[192];
        zenc_plan_block(hist.data(), bn, pl, len, tree);
        if ((size_t)(p - dst) + pl.csize > cap) return -1;
        u32 off = zenc_write_block_prefix(p, pl, tree, b + 1 == nblk, bn ? s[0] : 0);
        if (pl.kind == ZK_RAW) memcpy(p + off, s, bn);
        else if (pl.kind == ZK_HUF) {
            u16 code[256]; u32 codes[256];
            huf_assign_codes(len, pl.log, code);
            for (u32 i = 0; i < 256; i++) codes[i] = code[i] | ((u32)len[i] << 16);
            u32 o = off;
            for (u32 k = 0; k < 4; k++) {
                u32 cnt = k < 3 ? per : bn - 3 * per;
                u32 w = huf_encode_stream(p + o, s + k * per, cnt, codes);
                if (w != pl.ssz[k]) return -100 - (long long)k;
                o += w;
            }
            p[o] = 0;                                           // Number_of_Sequences = 0
            if (o + 1 != pl.csize) return -2;
        }
        p += pl.csize;
    }
    return p - dst;
}
