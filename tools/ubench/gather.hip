// gather.hip — how fast, and how FETCH_SIZE counts, random 64-B and 112-B row gathers from a table the size of the
// 1M-triangle scene's BVH (larger than the L2s, smaller than the 256 MiB Infinity Cache): the access pattern of
// wf_intersect_kernel (a 4-wide node = 7 float4 of a 128-B row, a leaf row = 4 float4 of a 64-B row, one row per lane
// per step, the next row known only when this one has arrived).  Two uses:
//   ./gather                      prints rows/s and bytes/s per configuration: the "L3 gather" comparator of the roofline
//   rocprofv3 --pmc FETCH_SIZE -- ./gather --one MODE   the known byte count of ONE launch to hold against the counter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int NV, int WAVES>
__global__ __launch_bounds__(256, WAVES) void gather_kernel(const float4* __restrict__ table, uint32_t n_rows, int row_f4, int iters,
                                                            int dependent, float* __restrict__ out)
{
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    uint32_t x = tid * 2654435761u + 12345u;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t r = __umulhi(x, n_rows);
        const float4* p = table + (size_t)r * (size_t)row_f4;
        float4 v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = p[k];
#pragma unroll
        for (int k = 0; k < NV; ++k) acc += v[k].x + v[k].w;
        if (dependent) x ^= __float_as_uint(acc) & 0xffu;      // the next row is known only when this one has arrived
    }
    out[tid] = acc;
}

// Cooperative form of the same gather (round 3): the lanes of a wave still want one row each, but the row of lane j is
// fetched by Q adjacent lanes, 16 bytes each (Q = 4 for a 64-B row, 8 for a 128-B row), so that one load instruction
// touches 64 / Q rows with Q lanes per cache line instead of 64 rows with one lane per line; Q instructions fetch all 64
// rows.  The address unit of a CU takes about one DIVERGENT 16-byte lane per clock; this asks whether lanes that share a
// line share the clock.  LDS = 1: every piece then goes through LDS (ds_write_b128 where its owner expects it, then the
// owner reads its row with NV ds_read_b128), which is what a traversal kernel would have to pay to get its row back.
template <int Q, int WAVES, int LDS>
__global__ __launch_bounds__(256, WAVES) void gather_coop_kernel(const float4* __restrict__ table, uint32_t n_rows, int iters,
                                                                 int dependent, float* __restrict__ out)
{
    __shared__ float4 stage[LDS ? 256 * Q : 1];
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const int lane = (int)(threadIdx.x & 63u), piece = lane & (Q - 1), sub = lane / Q;      // sub: which of the 64 / Q rows of an instruction
    float4* wave_stage = stage + (LDS ? (threadIdx.x & ~63u) * Q : 0);
    uint32_t x = tid * 2654435761u + 12345u;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t r = __umulhi(x, n_rows);                 // the row THIS lane wants
        float4 v[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            const int owner = k * (64 / Q) + sub;               // the lane whose row this instruction's group `sub` fetches
            const uint32_t ro = (uint32_t)__builtin_amdgcn_ds_bpermute(owner << 2, (int)r);
            v[k] = table[(size_t)ro * Q + piece];
        }
        if (LDS) {
#pragma unroll
            for (int k = 0; k < Q; ++k) wave_stage[(k * (64 / Q) + sub) * Q + piece] = v[k];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < Q; ++k) v[k] = wave_stage[lane * Q + k];
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int k = 0; k < Q; ++k) acc += v[k].x + v[k].w;
        if (dependent) x ^= __float_as_uint(acc) & 0xffu;
    }
    out[tid] = acc;
}

struct Mode { const char* name; int nv, row_bytes, waves, dependent; int coop = 0, lds = 0; };
static const Mode MODES[] = {
    {"leaf64_dep_5w", 4, 64, 5, 1}, {"node112_dep_5w", 7, 128, 5, 1}, {"leaf64_dep_8w", 4, 64, 8, 1}, {"node112_dep_8w", 7, 128, 8, 1},
    {"leaf64_indep_8w", 4, 64, 8, 0}, {"node112_indep_8w", 7, 128, 8, 0},
    {"coop64_dep_5w", 4, 64, 5, 1, 1, 0}, {"coop128_dep_5w", 8, 128, 5, 1, 1, 0},
    {"coop64_lds_dep_5w", 4, 64, 5, 1, 1, 1}, {"coop128_lds_dep_5w", 8, 128, 5, 1, 1, 1},
    {"coop64_dep_8w", 4, 64, 8, 1, 1, 0}, {"coop128_dep_8w", 8, 128, 8, 1, 1, 0},
};

static void launch(const Mode& m, const float4* table, uint32_t n_rows, int iters, float* out, int blocks)
{
    const int row_f4 = m.row_bytes / 16;
    if (m.coop) {
#define LC(Q, W, LD) gather_coop_kernel<Q, W, LD><<<blocks, 256>>>(table, n_rows, iters, m.dependent, out)
        if (m.nv == 4 && m.waves == 5 && !m.lds) LC(4, 5, 0); else if (m.nv == 8 && m.waves == 5 && !m.lds) LC(8, 5, 0);
        else if (m.nv == 4 && m.lds) LC(4, 5, 1); else if (m.nv == 8 && m.lds) LC(8, 5, 1);
        else if (m.nv == 4) LC(4, 8, 0); else LC(8, 8, 0);
#undef LC
        return;
    }
#define L(NV, W) gather_kernel<NV, W><<<blocks, 256>>>(table, n_rows, row_f4, iters, m.dependent, out)
    if (m.nv == 4 && m.waves == 5) L(4, 5); else if (m.nv == 7 && m.waves == 5) L(7, 5);
    else if (m.nv == 4) L(4, 8); else L(7, 8);
#undef L
}

int main(int argc, char** argv)
{
    size_t table_mb = 110;
    int only = -1, iters = 2000;
    for (int a = 1; a < argc; ++a) {
        if (!std::strcmp(argv[a], "--one") && a + 1 < argc) only = std::atoi(argv[++a]);
        else if (!std::strcmp(argv[a], "--mb") && a + 1 < argc) table_mb = (size_t)std::atoll(argv[++a]);
        else if (!std::strcmp(argv[a], "--iters") && a + 1 < argc) iters = std::atoi(argv[++a]);
    }
    const size_t bytes = table_mb << 20;
    float4* table = nullptr;
    CK(hipMalloc((void**)&table, bytes));
    std::vector<float> host(bytes / 4);
    uint32_t s = 1u;
    for (float& f : host) { s = s * 1664525u + 1013904223u; f = (float)(s >> 8) * (1.0f / 16777216.0f); }
    CK(hipMemcpy(table, host.data(), bytes, hipMemcpyHostToDevice));
    float* out = nullptr;
    const int max_blocks = 256 * 8;
    CK(hipMalloc((void**)&out, (size_t)max_blocks * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mi = 0; mi < (int)(sizeof MODES / sizeof MODES[0]); ++mi) {
        if (only >= 0 && mi != only) continue;
        const Mode& m = MODES[mi];
        const uint32_t n_rows = (uint32_t)(bytes / (size_t)m.row_bytes);
        const int blocks = 256 * m.waves;                       // 4 waves per block: `waves` blocks per CU fill `waves` per SIMD
        if (only < 0) { launch(m, table, n_rows, 50, out, blocks); CK(hipDeviceSynchronize()); }     // warm-up (not under the counter run)
        CK(hipEventRecord(e0));
        launch(m, table, n_rows, iters, out, blocks);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double rows = (double)blocks * 256.0 * iters, got = rows * m.nv * 16.0;
        std::printf("{\"mode\": \"%s\", \"table_MB\": %zu, \"row_bytes_read\": %d, \"row_stride\": %d, \"waves_per_simd\": %d, \"dependent\": %d, "
                    "\"rows\": %.0f, \"bytes_read\": %.0f, \"ms\": %.3f, \"Grows_per_s\": %.3f, \"TB_per_s\": %.3f}\n",
                    m.name, table_mb, m.nv * 16, m.row_bytes, m.waves, m.dependent, rows, got, ms, rows / ms / 1e6, got / ms / 1e9);
    }
    return 0;
}
