// What one LDS operation of a wavefront costs on this device, by kind: tools/lds_probe  (hipcc --offload-arch=gfx950 -O3 -o tools/lds_probe tools/lds_probe.hip)
// Every kernel runs ITER trips of UNR operations per lane on addresses that depend on a per-lane pseudo-random symbol (as a histogram's
// or a decoding table's do), WGS workgroups of 256 threads per CU.  Printed: device-wide operations per second and, from them,
// cycles of a CU's LDS per wavefront operation at the clock given on the command line (default 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef uint32_t u32; typedef uint8_t u8; typedef uint16_t u16;
#define ITER 2048
#define UNR 8
__device__ __forceinline__ u32 rnd(u32 &s) { s = s * 1664525u + 1013904223u; return s >> 24; }

// MODE 0: ds_read_u8 + ds_write_b8, lane-owned byte counters [sym][lane][wave]      (64 KiB)
// MODE 1: ds_read_b32 + ds_write_b32, lane-owned dword counters [sym & 63][thread]    (64 KiB)
// MODE 2: ds_add_u32 (no return), lane-owned dword [sym & 63][thread]                 (64 KiB)
// MODE 3: ds_add_u32 (no return), shared histogram of 256 dwords x 4 copies           (4 KiB)
// MODE 4: ds_read_u16 from a shared table of 256 entries (a decoding table)           (512 B)
// MODE 5: ds_read_b32 from a shared table of 256 entries                              (1 KiB)
// MODE 6: ds_read_u8 only, lane-owned bytes as MODE 0
// MODE 7: ds_write_b8 only, lane-owned bytes as MODE 0
template <int MODE>
__global__ __launch_bounds__(256) void k_probe(u32 *sink)
{
    __shared__ __attribute__((aligned(16))) u32 lds[MODE <= 2 || MODE >= 6 ? 16384 : 1024];
    const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (u32 i = t; i < (MODE <= 2 || MODE >= 6 ? 16384u : 1024u); i += 256) lds[i] = i;
    __syncthreads();
    u32 s = t * 2654435761u + blockIdx.x, acc = 0;
    u8 *const b8 = (u8 *)lds + lane * 4 + wave;
    for (u32 it = 0; it < ITER; it++) {
#pragma unroll
        for (u32 k = 0; k < UNR; k++) {
            const u32 sym = rnd(s);
            if (MODE == 0) { u8 *p = b8 + (sym << 8); *p = (u8)(*p + 1); }
            else if (MODE == 1) { u32 *p = lds + (sym & 63) * 256 + t; *p = *p + 1; }
            else if (MODE == 2) atomicAdd(lds + (sym & 63) * 256 + t, 1u);
            else if (MODE == 3) atomicAdd(lds + (t & 3) * 256 + sym, 1u);
            else if (MODE == 4) acc += ((const u16 *)lds)[sym];
            else if (MODE == 5) acc += lds[sym];
            else if (MODE == 6) acc += b8[sym << 8];
            else if (MODE == 7) b8[sym << 8] = (u8)it;
        }
    }
    __syncthreads();
    if (acc == 0x12345u || lds[t] == 0xFFFFFFFFu) sink[0] = acc;
}

template <int MODE> static void run(const char *name, int ops_per_sym, int wgs_per_cu, int cus, double ghz, u32 *sink)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = cus * wgs_per_cu;
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double syms = (double)grid * 256 * ITER * UNR;                      // per-lane operations groups
    const double waveops_per_cu = syms * ops_per_sym / 64 / cus;              // wavefront LDS instructions per CU
    const double cyc = ms * 1e-3 * ghz * 1e9;
    printf("%-44s wgs/cu %2d  %8.3f ms  %7.2f G lane-symbols/s  %6.1f cycles per wavefront LDS instruction per CU\n", name, wgs_per_cu, ms, syms / ms / 1e6, cyc / waveops_per_cu);
}

int main(int argc, char **argv)
{
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs, LDS per workgroup %zu, clock %d MHz\n", p.name, cus, (size_t)p.sharedMemPerBlock, p.clockRate / 1000);
    u32 *sink; hipMalloc(&sink, 64);
    for (int w = 1; w <= 2; w++) {
        run<0>("u8 read + u8 write, lane-owned", 2, w, cus, ghz, sink);
        run<6>("u8 read, lane-owned", 1, w, cus, ghz, sink);
        run<7>("u8 write, lane-owned", 1, w, cus, ghz, sink);
        run<1>("b32 read + b32 write, lane-owned", 2, w, cus, ghz, sink);
        run<2>("ds_add_u32, lane-owned", 1, w, cus, ghz, sink);
    }
    for (int w = 1; w <= 8; w *= 2) {
        run<3>("ds_add_u32, shared 256 x 4 copies", 1, w, cus, ghz, sink);
        run<4>("u16 read, shared table", 1, w, cus, ghz, sink);
        run<5>("b32 read, shared table", 1, w, cus, ghz, sink);
    }
    return 0;
}
