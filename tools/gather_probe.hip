// Stand-alone probe of the random-row gather ceiling of one GPU: out[r] = sum of DEG random rows of a table.
// Sweeps table size, row bytes and rows in flight per lane group.  Build: hipcc -O3 --offload-arch=gfx950
// tools/gather_probe.hip -o build/gather_probe ; it is a measurement tool, not part of the library.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            std::exit(1);                                                                \
        }                                                                                \
    } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ void k_fill_idx(int* idx, long n, unsigned rows_mask) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (int)(hash32((unsigned)i * 2654435761U + 12345U) & rows_mask);
}

// LPR lanes own one output row (16 B per lane); U gathers in flight per group; DEG indices per output row.
template <int LPR, int U, int WAVES, int MODE>
__global__ void __launch_bounds__(256, WAVES) k_gather_sum(const float4* __restrict__ table, const int* __restrict__ idx,
                                                           float4* __restrict__ out, long n_out, int deg,
                                                           const float4* __restrict__ ea, const float4* __restrict__ we) {
    const int lane = threadIdx.x & 63;
    const int lig = lane & (LPR - 1);
    const int gbase = lane & ~(LPR - 1);
    const long group = (blockIdx.x * (long)blockDim.x + threadIdx.x) / LPR;
    if (group >= n_out) return;
    const int* my = idx + group * deg;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    float4 w0 = we[lig * 4 + 0], w1 = we[lig * 4 + 1], w2 = we[lig * 4 + 2], w3 = we[lig * 4 + 3];
    const float4* myea = ea + group * deg;
    for (int e0 = 0; e0 < deg; e0 += LPR) {
        int mine = (e0 + lig < deg) ? my[e0 + lig] : 0;
        int lim = deg - e0 < LPR ? deg - e0 : LPR;
        float4 eam = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 2 && e0 + lig < deg) eam = myea[e0 + lig];
        for (int j = 0; j < lim; j += U) {
            float4 v[U], ev[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int jj = j + u < lim ? j + u : lim - 1;
                int row = __shfl(mine, gbase + jj);
                v[u] = table[(long)row * LPR + lig];
                if (MODE == 1) ev[u] = myea[e0 + jj];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (MODE == 2) {
                    int jj = j + u < lim ? j + u : lim - 1;
                    ev[u].x = __shfl(eam.x, gbase + jj); ev[u].y = __shfl(eam.y, gbase + jj);
                    ev[u].z = __shfl(eam.z, gbase + jj); ev[u].w = __shfl(eam.w, gbase + jj);
                }
                if (MODE != 0) {
                    float4 e = ev[u];
                    v[u].x = fmaxf(v[u].x + (w0.x * e.x + w0.y * e.y + w0.z * e.z + w0.w * e.w), 0.f);
                    v[u].y = fmaxf(v[u].y + (w1.x * e.x + w1.y * e.y + w1.z * e.z + w1.w * e.w), 0.f);
                    v[u].z = fmaxf(v[u].z + (w2.x * e.x + w2.y * e.y + w2.z * e.z + w2.w * e.w), 0.f);
                    v[u].w = fmaxf(v[u].w + (w3.x * e.x + w3.y * e.y + w3.z * e.z + w3.w * e.w), 0.f);
                }
                if (j + u < lim) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
        }
    }
    out[group * LPR + lig] = acc;
}

static const float4* g_ea;
static const float4* g_we;
template <int LPR, int U, int WAVES, int MODE = 0>
static float run(const float4* table, const int* idx, float4* out, long n_out, int deg) {
    long threads = n_out * LPR;
    int blocks = (int)((threads + 255) / 256);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    k_gather_sum<LPR, U, WAVES, MODE><<<blocks, 256>>>(table, idx, out, n_out, deg, g_ea, g_we);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 3; ++i) k_gather_sum<LPR, U, WAVES, MODE><<<blocks, 256>>>(table, idx, out, n_out, deg, g_ea, g_we);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 3;
}

int main(int argc, char** argv) {
    const bool contig = argc > 1 && argv[1][0] == 'c';   // physically contiguous table (hipDeviceMallocContiguous)
    const int deg = 25;
    const long n_out = 2'000'000;   // 50 M gathers per launch
    const long max_table_bytes = 4L << 30;
    float4* table;
    int* idx;
    float4* out;
    if (contig) CHECK(hipExtMallocWithFlags((void**)&table, max_table_bytes, hipDeviceMallocContiguous));
    else CHECK(hipMalloc(&table, max_table_bytes));
    std::printf("table allocation: %s\n", contig ? "hipDeviceMallocContiguous" : "hipMalloc");
    CHECK(hipMemset(table, 0, max_table_bytes));
    CHECK(hipMalloc(&idx, n_out * deg * sizeof(int)));
    CHECK(hipMalloc(&out, n_out * 512));
    float4 *ea, *we;
    CHECK(hipMalloc(&ea, n_out * deg * sizeof(float4)));
    CHECK(hipMemset(ea, 0, n_out * deg * sizeof(float4)));
    CHECK(hipMalloc(&we, 128 * sizeof(float4)));
    CHECK(hipMemset(we, 0, 128 * sizeof(float4)));
    g_ea = ea; g_we = we;
    for (int rowb : {512, 256}) {
        for (int lg = 20; lg <= 23; ++lg) {
            long rows = 1L << lg;
            if (rows * rowb > max_table_bytes) continue;
            k_fill_idx<<<(int)((n_out * deg + 255) / 256), 256>>>(idx, n_out * deg, (unsigned)(rows - 1));
            CHECK(hipDeviceSynchronize());
            float t[6];
            if (rowb == 512) {
                t[0] = run<32, 4, 4>(table, idx, out, n_out, deg);
                t[1] = run<32, 4, 4, 1>(table, idx, out, n_out, deg);
                t[2] = run<32, 4, 4, 2>(table, idx, out, n_out, deg);
                t[3] = run<32, 8, 4, 1>(table, idx, out, n_out, deg);
                t[4] = run<32, 8, 4, 2>(table, idx, out, n_out, deg);
                t[5] = run<32, 2, 4, 2>(table, idx, out, n_out, deg);
            } else {
                t[0] = run<16, 4, 4>(table, idx, out, n_out, deg);
                t[1] = run<16, 4, 4, 1>(table, idx, out, n_out, deg);
                t[2] = run<16, 4, 4, 2>(table, idx, out, n_out, deg);
                t[3] = run<16, 8, 4, 1>(table, idx, out, n_out, deg);
                t[4] = run<16, 8, 4, 2>(table, idx, out, n_out, deg);
                t[5] = run<16, 2, 4, 2>(table, idx, out, n_out, deg);
            }
            std::printf("row %3d B  table %6ld MB :", rowb, rows * rowb >> 20);
            const char* names[6] = {"U4pure", "U4bcast", "U4shfl", "U8bcast", "U8shfl", "U2shfl"};
            for (int i = 0; i < 6; ++i)
                std::printf("  %s %5.2f Grow/s %4.0f GB/s", names[i], n_out * deg / t[i] / 1e6,
                            n_out * deg * (double)rowb / t[i] / 1e6);
            std::printf("\n");
            std::fflush(stdout);
        }
    }
    return 0;
}
