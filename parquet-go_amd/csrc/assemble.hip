// assemble.hip — level post-processing on the device (gfx950).
//
// 1. levels_pack: the bytes of the reference's PackedArray for a run of levels
//    (pack8int32_w, encoding/bitpacking32.go:45-79; container: encoding/packed-array.go:47-55,
//    118-125): value i of a group sits in bits [i w, (i + 1) w) of the group's w bytes.
// 2. column entries: what ColumnStore.Get (datastore/column.go:127-177) yields call by call over
//    a column's definition / repetition levels, as arrays: per entry (= call) a validity bit, the
//    definition level returned, and the index of its first value.
//
//    With c[i] = REPEATED && rep[i] >= maxR and d[i] = def[i] >= maxD, "position i belongs to an
//    entry that returned a value" obeys  V[i] = d[i] | (c[i] & V[i-1])  — the carry chain of a
//    binary addition with generate d and propagate c & ~d.  Over the 64 positions of a wave step
//    the chain is ONE 64-bit scalar addition of the two ballots:  A = C | D,  S = A + D + V_in,
//    carries into the bits X = S ^ A ^ D,  V = X >> 1 | carry_out << 63.  Position i opens an
//    entry unless c[i] & V[i-1]; an entry takes a value at every position with V[i].
//    Three launches: per tile of 2048 positions the summary {V_out, entries, values} for both
//    possible V_in; one workgroup chains the tiles; the tiles then write their entries.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "pq_device.hpp"
#include <mutex>

#include "pq_launch.hpp"

namespace pq {

// orders this wave's LDS traffic for the compiler (DS operations of one wave execute in order)
static __device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

static __device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// x of lane `l` (wave-uniform), as a scalar
static __device__ __forceinline__ uint64_t readlane_u64(uint64_t x, uint32_t l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, (int)l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), (int)l);
    return ((uint64_t)hi << 32) | lo;
}

constexpr int kAsmSteps = 32;                    // wave steps per tile
constexpr int kAsmTile = kAsmSteps * kWave;      // positions per tile (one wave)
constexpr int kAsmWG = 256;                      // 4 tiles per workgroup
constexpr int kAsmScanWG = 1024;
constexpr int kAsmVbWords = 576;                 // LDS dwords per tile for validity bytes (>= (2048 + 32 + 31) / 4)

// V chain over one step.  x: bit i = V[i-1] (the carry into position i); v: bit i = V[i].
__device__ __forceinline__ void asm_step(uint64_t C, uint64_t D, uint32_t vin, uint64_t &x, uint64_t &v, uint32_t &vout)
{
    const uint64_t a = C | D;
    const unsigned __int128 s = (unsigned __int128)a + D + vin;
    const uint64_t lo = (uint64_t)s;
    vout = (uint32_t)(s >> 64);
    x = lo ^ a ^ D;
    v = (x >> 1) | ((uint64_t)vout << 63);
}

// The steps of a tile as a chain of their own: lane s holds the ballots of step s (zero past the
// last step).  g: steps that let a V out whatever comes in, a: steps that let one out when one
// comes in (g included).
struct BlockChain {
    uint64_t g, a;
};

__device__ __forceinline__ BlockChain asm_block_chain(uint64_t my_c, uint64_t my_d)
{
    uint64_t x, v;
    uint32_t out0, out1;
    asm_step(my_c, my_d, 0u, x, v, out0);
    asm_step(my_c, my_d, 1u, x, v, out1);
    BlockChain bc;
    bc.g = __builtin_amdgcn_ballot_w64(out0 != 0u);
    bc.a = __builtin_amdgcn_ballot_w64(out1 != 0u);
    return bc;
}

// Bit s = the V that enters step s when `vin` enters the tile (bit `steps` = the V that leaves it).
__device__ __forceinline__ uint64_t asm_chain_in(const BlockChain &bc, uint32_t vin)
{
    return (bc.a + bc.g + vin) ^ bc.a ^ bc.g;
}

// Ballots of one step: C, D and the positions that exist.
struct StepMasks {
    uint64_t c, d, valid;
};

__device__ __forceinline__ StepMasks asm_masks(const int32_t *__restrict__ def, const int32_t *__restrict__ rep,
                                               uint64_t pos, uint64_t n, int32_t max_def, int32_t max_rep,
                                               bool repeated)
{
    const bool in = pos < n;
    const int32_t dl = in ? def[pos] : 0;
    const int32_t rl = (in && repeated) ? rep[pos] : 0;
    StepMasks m;
    m.valid = __builtin_amdgcn_ballot_w64(in);
    m.d = __builtin_amdgcn_ballot_w64(in && dl >= max_def);
    m.c = __builtin_amdgcn_ballot_w64(in && repeated && rl >= max_rep);
    return m;
}

// Tile summaries: for V_in = 0 and V_in = 1 the V that leaves the tile, the entries it opens and
// the values it takes (tile_sum[8 t + 0..4] = {V_out(0) | V_out(1) << 1, entries(0), entries(1),
// values(0), values(1)}), the same for the workgroup's four tiles together (group_sum), and the
// ballots (16 bytes per step) for the last pass.
__global__ __launch_bounds__(kAsmWG) void entries_summary_kernel(const int32_t *__restrict__ def,
                                                                 const int32_t *__restrict__ rep, uint64_t n,
                                                                 int32_t max_def, int32_t max_rep, int repeated,
                                                                 uint64_t *__restrict__ masks,
                                                                 uint32_t *__restrict__ tile_sum,
                                                                 uint32_t *__restrict__ group_sum)
{
    __shared__ uint32_t wsum[kAsmWG / kWave][5];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const uint64_t tile = (uint64_t)blockIdx.x * (kAsmWG / kWave) + wave;
    const uint64_t base = tile * kAsmTile;
    const uint32_t steps = base >= n ? 0u : (uint32_t)min((uint64_t)kAsmSteps, (n - base + kWave - 1) / kWave);

    // ballots first (vector loads, nothing serial), lane s keeps those of step s
    uint64_t my_c = 0, my_d = 0, my_valid = 0;
    if (steps == (uint32_t)kAsmSteps && base + kAsmTile <= n) {
        // a whole tile: no bounds, sixteen steps' loads in flight
        const int32_t *dp = def + base + lane;
        const int32_t *rp = repeated ? rep + base + lane : dp;
        my_valid = lane < kAsmSteps ? ~0ull : 0ull;  // lane s stands for step s; there are no steps past the tile
#pragma unroll
        for (uint32_t s0 = 0; s0 < (uint32_t)kAsmSteps; s0 += 16) {
            int32_t dl[16], rl[16];
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) dl[j] = dp[(s0 + j) * kWave];
            if (repeated) {
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) rl[j] = rp[(s0 + j) * kWave];
            }
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) {
                const uint64_t D = __builtin_amdgcn_ballot_w64(dl[j] >= max_def);
                const uint64_t C = repeated ? __builtin_amdgcn_ballot_w64(rl[j] >= max_rep) : 0ull;
                if ((uint32_t)lane == s0 + j) {
                    my_c = C;
                    my_d = D;
                }
            }
        }
    } else {
        for (uint32_t s0 = 0; s0 < steps; s0 += 8) {
            StepMasks m[8];  // eight steps' loads in flight (positions past n load nothing)
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j)
                m[j] = asm_masks(def, rep, base + (uint64_t)(s0 + j) * kWave + lane, n, max_def, max_rep, repeated != 0);
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j)
                if ((uint32_t)lane == s0 + j) {
                    my_c = m[j].c;
                    my_d = m[j].d;
                    my_valid = m[j].valid;
                }
        }
    }
    if ((uint32_t)lane < steps) {
        masks[2 * (tile * kAsmSteps + lane)] = my_c;
        masks[2 * (tile * kAsmSteps + lane) + 1] = my_d;
    }

    // every lane takes its own step through both chains: first the V that would leave it for
    // either V_in, which makes the steps themselves a carry chain (one more scalar addition per
    // tile V_in), then its counts under the V that really enters it
    const BlockChain bc = asm_block_chain(my_c, my_d);
    uint32_t hs[2], ns[2], vouts[2];
#pragma unroll
    for (uint32_t k = 0; k < 2; ++k) {
        const uint64_t xb = asm_chain_in(bc, k);
        uint64_t x, v;
        uint32_t vo;
        asm_step(my_c, my_d, (uint32_t)(xb >> lane) & 1u, x, v, vo);
        hs[k] = wave_sum_u32((uint32_t)__builtin_popcountll(~(my_c & x) & my_valid));
        ns[k] = wave_sum_u32((uint32_t)__builtin_popcountll(v & my_valid));
        vouts[k] = (uint32_t)(xb >> steps) & 1u;
    }
    const uint32_t v0 = vouts[0], v1 = vouts[1], h0 = hs[0], h1 = hs[1], n0 = ns[0], n1 = ns[1];
    if (lane == 0) {
        wsum[wave][0] = v0 | (v1 << 1);
        wsum[wave][1] = h0;
        wsum[wave][2] = h1;
        wsum[wave][3] = n0;
        wsum[wave][4] = n1;
        if (steps != 0u) {
            uint32_t *t = tile_sum + 8 * tile;
#pragma unroll
            for (int k = 0; k < 5; ++k) t[k] = wsum[wave][k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t v[2] = {0, 1}, h[2] = {0, 0}, c[2] = {0, 0};
        for (int w = 0; w < kAsmWG / kWave; ++w)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint32_t vi = v[k];
                h[k] += wsum[w][1 + vi];
                c[k] += wsum[w][3 + vi];
                v[k] = (wsum[w][0] >> vi) & 1u;
            }
        uint32_t *g = group_sum + 8 * (uint64_t)blockIdx.x;
        g[0] = v[0] | (v[1] << 1);
        g[1] = h[0];
        g[2] = h[1];
        g[3] = c[0];
        g[4] = c[1];
    }
}

// A run of summaries as one function of the V that enters it.
struct AsmFn {
    uint32_t v;     // bit k: V that leaves for V_in = k
    uint32_t h[2];  // entries opened
    uint32_t c[2];  // values taken
};

__device__ __forceinline__ AsmFn asm_identity()
{
    AsmFn f;
    f.v = 2u;
    f.h[0] = f.h[1] = f.c[0] = f.c[1] = 0u;
    return f;
}

// f followed by g
__device__ __forceinline__ AsmFn asm_then(const AsmFn &f, uint32_t gv, uint32_t gh0, uint32_t gh1, uint32_t gc0,
                                          uint32_t gc1)
{
    AsmFn r;
    r.v = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t mid = (f.v >> k) & 1u;
        r.h[k] = f.h[k] + (mid ? gh1 : gh0);
        r.c[k] = f.c[k] + (mid ? gc1 : gc0);
        r.v |= ((gv >> mid) & 1u) << k;
    }
    return r;
}

// One workgroup chains the group summaries: group_sum[8 g + 5..7] = {V_in, entries before, values
// before} of group g; *result = totals; offsets[entries] = values.  Three levels: every thread
// folds its share of the groups, 32 threads fold 32 of those each, one thread folds the 32.
__global__ __launch_bounds__(kAsmScanWG) void entries_scan_kernel(uint32_t *__restrict__ group_sum, uint64_t groups,
                                                                  uint32_t *__restrict__ offsets,
                                                                  pq_entries_result *__restrict__ result)
{
    __shared__ uint32_t f1[5][kAsmScanWG];   // per thread
    __shared__ uint32_t f2[5][32];           // per 32 threads
    __shared__ uint32_t in1[3][kAsmScanWG];  // state entering a thread's share
    __shared__ uint32_t in2[3][32];
    const uint32_t t = threadIdx.x;
    const uint64_t per = (groups + kAsmScanWG - 1) / kAsmScanWG;
    const uint64_t lo = min(groups, (uint64_t)t * per), hi = min(groups, lo + per);
    // a thread's share stays in registers when it is small (up to 16 K groups, 134 M positions)
    constexpr int kKeep = 16;
    const bool keep = per <= (uint64_t)kKeep;
    uint4 q[kKeep];
    uint32_t q4[kKeep];
    AsmFn f = asm_identity();
    if (keep) {
#pragma unroll
        for (int j = 0; j < kKeep; ++j) {
            q[j] = make_uint4(2u, 0u, 0u, 0u);  // identity
            q4[j] = 0u;
            if (lo + j < hi) {
                q[j] = *reinterpret_cast<const uint4 *>(group_sum + 8 * (lo + j));
                q4[j] = group_sum[8 * (lo + j) + 4];
            }
        }
#pragma unroll
        for (int j = 0; j < kKeep; ++j) f = asm_then(f, q[j].x, q[j].y, q[j].z, q[j].w, q4[j]);
    } else {
        for (uint64_t i = lo; i < hi; ++i) {
            const uint4 g = *reinterpret_cast<const uint4 *>(group_sum + 8 * i);
            f = asm_then(f, g.x, g.y, g.z, g.w, group_sum[8 * i + 4]);
        }
    }
    f1[0][t] = f.v;
    f1[1][t] = f.h[0];
    f1[2][t] = f.h[1];
    f1[3][t] = f.c[0];
    f1[4][t] = f.c[1];
    __syncthreads();
    if (t < 32) {
        AsmFn g = asm_identity();
        for (uint32_t r = 32 * t; r < 32 * t + 32; ++r) g = asm_then(g, f1[0][r], f1[1][r], f1[2][r], f1[3][r], f1[4][r]);
        f2[0][t] = g.v;
        f2[1][t] = g.h[0];
        f2[2][t] = g.h[1];
        f2[3][t] = g.c[0];
        f2[4][t] = g.c[1];
    }
    __syncthreads();
    if (t == 0) {
        uint32_t cv = 0, ch = 0, cn = 0;
        for (int r = 0; r < 32; ++r) {
            in2[0][r] = cv;
            in2[1][r] = ch;
            in2[2][r] = cn;
            ch += f2[1 + cv][r];
            cn += f2[3 + cv][r];
            cv = (f2[0][r] >> cv) & 1u;
        }
        result->status = PQ_OK;
        result->n_entries = ch;
        result->n_values = cn;
        result->reserved = 0;
        offsets[ch] = cn;
    }
    __syncthreads();
    if (t < 32) {
        uint32_t cv = in2[0][t], ch = in2[1][t], cn = in2[2][t];
        for (uint32_t r = 32 * t; r < 32 * t + 32; ++r) {
            in1[0][r] = cv;
            in1[1][r] = ch;
            in1[2][r] = cn;
            ch += f1[1 + cv][r];
            cn += f1[3 + cv][r];
            cv = (f1[0][r] >> cv) & 1u;
        }
    }
    __syncthreads();
    uint32_t cv = in1[0][t], ch = in1[1][t], cn = in1[2][t];
    if (keep) {
#pragma unroll
        for (int j = 0; j < kKeep; ++j) {
            if (lo + j < hi) {
                uint32_t *s = group_sum + 8 * (lo + j);
                s[5] = cv;
                s[6] = ch;
                s[7] = cn;
            }
            ch += cv ? q[j].z : q[j].y;
            cn += cv ? q4[j] : q[j].w;
            cv = (q[j].x >> cv) & 1u;
        }
        return;
    }
    for (uint64_t i = lo; i < hi; ++i) {
        uint32_t *s = group_sum + 8 * i;
        const uint4 g = *reinterpret_cast<const uint4 *>(s);
        const uint32_t c1 = s[4];
        s[5] = cv;
        s[6] = ch;
        s[7] = cn;
        ch += cv ? g.z : g.y;
        cn += cv ? c1 : g.w;
        cv = (g.x >> cv) & 1u;
    }
}

// Every tile writes its entries.  validity must be zero on entry: words shared between two
// tiles are or-ed in.
__global__ __launch_bounds__(kAsmWG) void entries_write_kernel(const int32_t *__restrict__ def, uint64_t n,
                                                               int32_t max_def, const uint64_t *__restrict__ masks,
                                                               const uint32_t *__restrict__ tile_sum,
                                                               const uint32_t *__restrict__ group_sum,
                                                               uint32_t *__restrict__ offsets,
                                                               uint32_t *__restrict__ validity,
                                                               int32_t *__restrict__ entry_def)
{
    __shared__ __attribute__((aligned(16))) uint32_t vbytes[kAsmWG / kWave][kAsmVbWords];
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint64_t tile = (uint64_t)blockIdx.x * (kAsmWG / kWave) + wave;
    const uint64_t base = tile * kAsmTile;
    if (base >= n) return;
    const uint32_t steps = (uint32_t)min((uint64_t)kAsmSteps, (n - base + kWave - 1) / kWave);
    uint64_t my_c = 0, my_d = 0;
    if ((uint32_t)lane < steps) {
        my_c = masks[2 * (tile * kAsmSteps + lane)];
        my_d = masks[2 * (tile * kAsmSteps + lane) + 1];
    }
    // the state that enters the workgroup, taken through the tiles in front of this one
    const uint32_t *gs = group_sum + 8 * (uint64_t)blockIdx.x;
    uint32_t vin = uni(gs[5]), K = uni(gs[6]), vb = uni(gs[7]);
    for (uint32_t w = 0; w < wave; ++w) {
        const uint32_t *ts = tile_sum + 8 * (tile - wave + w);
        K += uni(ts[1 + vin]);
        vb += uni(ts[3 + vin]);
        vin = (uni(ts[0]) >> vin) & 1u;
    }

    // every lane works out its own step: the V that enters it, then its heads and values
    const BlockChain bc = asm_block_chain(my_c, my_d);
    const uint64_t xb = asm_chain_in(bc, vin);
    // positions of this lane's step that exist (none for the lanes past the tile's last step)
    const uint64_t left = (uint32_t)lane < steps ? n - base - (uint64_t)lane * kWave : 0ull;
    const uint64_t my_valid = left >= 64 ? ~0ull : (1ull << left) - 1ull;
    uint64_t mx, my_v;
    uint32_t vo;
    asm_step(my_c, my_d, (uint32_t)(xb >> lane) & 1u, mx, my_v, vo);
    const uint64_t my_h = ~(my_c & mx) & my_valid;
    my_v &= my_valid;
    const uint32_t hcnt = (uint32_t)__builtin_popcountll(my_h), vcnt = (uint32_t)__builtin_popcountll(my_v);
    const uint32_t hincl = wave_scan_u32(hcnt), vincl = wave_scan_u32(vcnt);
    const uint32_t my_k = K + hincl - hcnt, my_vb = vb + vincl - vcnt;  // entries / values before this step
    const uint32_t tile_entries = wave_last_u32(hincl);
    // steps with an entry that has no value: it returns its own definition level
    const uint32_t need = (uint32_t)__builtin_amdgcn_ballot_w64((my_h & ~my_v) != 0ull);

    // validity: one byte per entry in LDS at (entry - first entry of the word K sits in), packed below
    uint32_t *vb32 = vbytes[wave];
    uint8_t *vb8 = reinterpret_cast<uint8_t *>(vb32);
#pragma unroll
    for (int k = 0; k < kAsmVbWords / kWave; ++k) vb32[lane + kWave * k] = 0;
    wave_fence();
    const uint32_t shift = K & 31u;

    for (uint32_t s0 = 0; s0 < steps; s0 += 8) {
        int32_t dls[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            dls[j] = max_def;
            const uint64_t pos = base + (uint64_t)(s0 + j) * kWave + lane;
            if (((need >> (s0 + j)) & 1u) != 0u && pos < n) dls[j] = def[pos];
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t s = s0 + j;
            if (s >= steps) break;
            const uint64_t H = readlane_u64(my_h, s), v = readlane_u64(my_v, s);
            const uint32_t Ks = (uint32_t)__builtin_amdgcn_readlane((int)my_k, (int)s);
            const uint32_t vbs = (uint32_t)__builtin_amdgcn_readlane((int)my_vb, (int)s);
            const bool head = ((H >> lane) & 1ull) != 0ull;
            const uint32_t vbit = (uint32_t)(v >> lane) & 1u;
            const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(H >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)H, 0u));
            const uint32_t vr = __builtin_amdgcn_mbcnt_hi((uint32_t)(v >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)v, 0u));
            // entries never outnumber positions: the bound only keeps a damaged summary from
            // turning into a store outside the caller's arrays
            if (head && (uint64_t)Ks + r < n) {
                offsets[Ks + r] = vbs + vr;
                entry_def[Ks + r] = vbit ? max_def : dls[j];
                vb8[shift + (Ks - K) + r] = (uint8_t)vbit;
            }
        }
    }
    wave_fence();
    // 32 bytes -> one word per lane (and lane 0 once more for the 65th); words this tile does not
    // own alone are or-ed in
    const uint32_t total_bits = shift + min(tile_entries, (uint32_t)kAsmTile);
    const uint32_t w0 = K >> 5;
    const uint64_t words_cap = (n + 31) / 32;  // same kind of bound as on the entry stores above
    const uint32_t nwords = w0 >= words_cap ? 0u : (uint32_t)min((uint64_t)((total_bits + 31u) >> 5), words_cap - w0);
    for (uint32_t wi = (uint32_t)lane; wi < nwords; wi += kWave) {
        uint32_t d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = vb32[8 * wi + k];
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) word |= (((d[k] * 0x01020408u) >> 24) & 0xfu) << (4 * k);
        const bool whole = (wi != 0u || shift == 0u) && 32u * wi + 32u <= total_bits;
        if (whole)
            validity[w0 + wi] = word;
        else if (word != 0u)
            atomicOr(validity + w0 + wi, word);
    }
}

// ---- the same in ONE launch ---------------------------------------------------------------------
// Summary, chaining and writing in one pass over the levels: a workgroup takes its four tiles'
// ballots (kept in registers), publishes the group's summary, finds the state that enters the
// group by looking back over the groups in front of it, publishes the state that leaves it, and
// writes its entries.  A group's two records are one 64-bit word each, payload and presence bit
// together, stored and loaded with relaxed agent-scope atomics: no fence, nothing to order.
//   agg[g]: bit 0 present | V_out(0), V_out(1) | entries(0), entries(1), values(0), values(1), 14 bits each
//   pre[g]: bit 0 present | V that leaves the group | entries in front of the next group, 30 bits | values, 30 bits
// Workgroups start in the order of their index, so the groups a look-back waits for are running
// or done.  n < 2^30 (the 30-bit counts); agg and pre must be zero on entry.
constexpr uint64_t kOnePassMaxN = 1ull << 30;
constexpr int kOnePassWG = 256;  // 4 tiles = 8192 positions per group: a group's counts fit 14 bits

__device__ __forceinline__ uint64_t agg_pack(uint32_t v, uint32_t h0, uint32_t h1, uint32_t c0, uint32_t c1)
{
    return 1ull | ((uint64_t)v << 1) | ((uint64_t)h0 << 3) | ((uint64_t)h1 << 17) | ((uint64_t)c0 << 31) | ((uint64_t)c1 << 45);
}

__device__ __forceinline__ uint64_t pre_pack(uint32_t v, uint32_t k, uint32_t vb)
{
    return 1ull | ((uint64_t)v << 1) | ((uint64_t)k << 2) | ((uint64_t)vb << 32);
}

// The state {V, entries, values} that enters group g.  Called by one whole wave.  Lane i looks at
// group j - i; the nearest group whose leaving state is known ends the search, the summaries of
// the groups between it and g are chained as the steps of a tile are: their V bits are a carry
// chain (one addition), and each then adds its counts under the V that enters it.
__device__ __forceinline__ void entries_look_back(const uint64_t *agg, const uint64_t *pre, uint64_t g, int lane,
                                                  uint32_t &vin, uint32_t &K, uint32_t &vb)
{
    AsmFn behind = asm_identity();  // the groups already passed over (nearer to g), as one function
    int64_t j = (int64_t)g - 1;
    while (true) {
        const int64_t idx = j - lane;
        uint64_t a = 0, p = 0;
        uint64_t known, waiting;
        // Lanes look only as far back as it takes: 8 groups at first, all 64 when none of those
        // knows its leaving state (every look is a memory transaction per lane that no cache serves).
        uint32_t width = 8;
        do {
            if (idx < 0) {
                p = pre_pack(0u, 0u, 0u);  // in front of the first group: nothing
            } else if ((uint32_t)lane < width) {
                if ((p & 1ull) == 0ull) p = __hip_atomic_load(pre + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (((p | a) & 1ull) == 0ull) a = __hip_atomic_load(agg + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            known = __builtin_amdgcn_ballot_w64((p & 1ull) != 0ull);
            // every group nearer than the nearest known state has to show its summary
            const uint64_t nearer = known ? ((known & (0ull - known)) - 1ull) : ~0ull;
            waiting = __builtin_amdgcn_ballot_w64(((p | a) & 1ull) == 0ull) & nearer;
            if (known == 0ull && width < (uint32_t)kWave && (waiting & ((1ull << width) - 1ull)) == 0ull)
                width = kWave;  // the near ones have all shown their summaries and none its state
            else if (waiting != 0ull)
                __builtin_amdgcn_s_sleep(1);
        } while (waiting != 0ull);
        const uint32_t fp = known ? (uint32_t)__builtin_ctzll(known) : (uint32_t)kWave;  // lanes below fp: summaries
        const bool mine = (uint32_t)lane < fp;
        const uint32_t av = mine ? (uint32_t)(a >> 1) & 3u : 2u;  // identity elsewhere
        const uint32_t h0 = mine ? (uint32_t)(a >> 3) & 0x3fffu : 0u, h1 = mine ? (uint32_t)(a >> 17) & 0x3fffu : 0u;
        const uint32_t c0 = mine ? (uint32_t)(a >> 31) & 0x3fffu : 0u, c1 = mine ? (uint32_t)(a >> 45) & 0x3fffu : 0u;
        // chain order is from the farthest group (highest lane) to the nearest (lane 0): reverse the ballots
        const uint64_t G = __builtin_bitreverse64(__builtin_amdgcn_ballot_w64((av & 1u) != 0u));
        const uint64_t A = __builtin_bitreverse64(__builtin_amdgcn_ballot_w64((av & 2u) != 0u));
        const uint32_t bit = 63u - (uint32_t)lane;  // this lane's place in the chain
        if (known != 0ull) {
            const uint64_t pw = readlane_u64(p, fp);
            const uint32_t v0 = (uint32_t)(pw >> 1) & 1u;
            const uint64_t x = (A + G + v0) ^ A ^ G;  // bit b: the V that enters chain place b (lanes >= fp are identities)
            const unsigned __int128 s = (unsigned __int128)A + G + v0;
            const uint32_t vout = (uint32_t)(s >> 64);
            const uint32_t my_in = (uint32_t)(x >> bit) & 1u;
            const uint32_t hs = wave_sum_u32(my_in ? h1 : h0), cs = wave_sum_u32(my_in ? c1 : c0);
            uint32_t v = vout, k = ((uint32_t)(pw >> 2) & 0x3fffffffu) + hs, c = ((uint32_t)(pw >> 32) & 0x3fffffffu) + cs;
            // then the groups passed over in earlier rounds
            k += behind.h[v];
            c += behind.c[v];
            v = (behind.v >> v) & 1u;
            vin = v;
            K = k;
            vb = c;
            return;
        }
        // no known state among these 64: they become one function, in front of `behind`
        AsmFn r;
        r.v = 0;
#pragma unroll
        for (uint32_t k = 0; k < 2u; ++k) {
            const uint64_t x = (A + G + k) ^ A ^ G;
            const unsigned __int128 s = (unsigned __int128)A + G + k;
            const uint32_t my_in = (uint32_t)(x >> bit) & 1u;
            r.h[k] = wave_sum_u32(my_in ? h1 : h0);
            r.c[k] = wave_sum_u32(my_in ? c1 : c0);
            r.v |= (uint32_t)(s >> 64) << k;
        }
        behind = asm_then(r, behind.v, behind.h[0], behind.h[1], behind.c[0], behind.c[1]);
        j -= kWave;
    }
}

template <int WG>
__global__ __launch_bounds__(WG) void entries_onepass_kernel(const int32_t *__restrict__ def,
                                                                 const int32_t *__restrict__ rep, uint64_t n,
                                                                 int32_t max_def, int32_t max_rep, int repeated,
                                                                 uint64_t *agg, uint64_t *pre, uint64_t groups,
                                                                 uint32_t *__restrict__ offsets,
                                                                 uint32_t *__restrict__ validity,
                                                                 int32_t *__restrict__ entry_def,
                                                                 pq_entries_result *__restrict__ result)
{
    __shared__ uint32_t wsum[WG / kWave][5];
    __shared__ uint32_t gstate[3];  // V, entries, values that enter the group
    __shared__ __attribute__((aligned(16))) uint32_t vbytes[WG / kWave][kAsmVbWords];
    const int lane = threadIdx.x & (kWave - 1);
    const uint32_t wave = uni(threadIdx.x >> 6);
    const uint64_t tile = (uint64_t)blockIdx.x * (WG / kWave) + wave;
    const uint64_t base = tile * kAsmTile;
    const uint32_t steps = base >= n ? 0u : (uint32_t)min((uint64_t)kAsmSteps, (n - base + kWave - 1) / kWave);

    // ---- ballots: lane s keeps those of step s ----
    uint64_t my_c = 0, my_d = 0;
    // The definition levels are wanted once more, for the entries without a value.  A whole tile
    // keeps them, a byte each (step s in byte s & 3 of dkeep[s >> 2]), unless one of them does
    // not fit a byte; the tile then reads them again, like the tile at the end of the column.
    uint32_t dkeep[kAsmSteps / 4];
#pragma unroll
    for (int k = 0; k < kAsmSteps / 4; ++k) dkeep[k] = 0;
    uint32_t dbits = 0xffffffffu;  // all the levels of this lane or-ed together
    if (steps == (uint32_t)kAsmSteps && base + kAsmTile <= n) {
        dbits = 0;
        const int32_t *dp = def + base + lane;
        const int32_t *rp = repeated ? rep + base + lane : dp;
#pragma unroll
        for (uint32_t s0 = 0; s0 < (uint32_t)kAsmSteps; s0 += 16) {
            int32_t dl[16], rl[16];
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) dl[j] = __builtin_nontemporal_load(dp + (s0 + j) * kWave);  // read once
            if (repeated) {
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) rl[j] = __builtin_nontemporal_load(rp + (s0 + j) * kWave);
            }
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) {
                const uint64_t D = __builtin_amdgcn_ballot_w64(dl[j] >= max_def);
                const uint64_t C = repeated ? __builtin_amdgcn_ballot_w64(rl[j] >= max_rep) : 0ull;
                if ((uint32_t)lane == s0 + j) {
                    my_c = C;
                    my_d = D;
                }
                dbits |= (uint32_t)dl[j];
                dkeep[(s0 + j) >> 2] |= ((uint32_t)dl[j] & 0xffu) << (8u * ((s0 + j) & 3u));
            }
        }
    } else {
        for (uint32_t s0 = 0; s0 < steps; s0 += 8) {
            StepMasks m[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j)
                m[j] = asm_masks(def, rep, base + (uint64_t)(s0 + j) * kWave + lane, n, max_def, max_rep, repeated != 0);
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j)
                if ((uint32_t)lane == s0 + j) {
                    my_c = m[j].c;
                    my_d = m[j].d;
                }
        }
    }
    // positions of this lane's step that exist (none for the lanes past the tile's last step)
    const uint64_t left = (uint32_t)lane < steps ? n - base - (uint64_t)lane * kWave : 0ull;
    const uint64_t my_valid = left >= 64 ? ~0ull : (1ull << left) - 1ull;

    // ---- the tile's summary for either V that may enter it ----
    const BlockChain bc = asm_block_chain(my_c, my_d);
    {
        uint32_t hs[2], ns[2], vouts[2];
#pragma unroll
        for (uint32_t k = 0; k < 2; ++k) {
            const uint64_t xb = asm_chain_in(bc, k);
            uint64_t x, v;
            uint32_t vo;
            asm_step(my_c, my_d, (uint32_t)(xb >> lane) & 1u, x, v, vo);
            hs[k] = wave_sum_u32((uint32_t)__builtin_popcountll(~(my_c & x) & my_valid));
            ns[k] = wave_sum_u32((uint32_t)__builtin_popcountll(v & my_valid));
            vouts[k] = (uint32_t)(xb >> steps) & 1u;
        }
        if (lane == 0) {
            wsum[wave][0] = vouts[0] | (vouts[1] << 1);
            wsum[wave][1] = hs[0];
            wsum[wave][2] = hs[1];
            wsum[wave][3] = ns[0];
            wsum[wave][4] = ns[1];
        }
    }
    __syncthreads();

    // ---- the group's summary out, the state that enters it in ----
    if (wave == 0u) {
        uint32_t v[2] = {0, 1}, h[2] = {0, 0}, c[2] = {0, 0};
        for (int w = 0; w < WG / kWave; ++w)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint32_t vi = v[k];
                h[k] += wsum[w][1 + vi];
                c[k] += wsum[w][3 + vi];
                v[k] = (wsum[w][0] >> vi) & 1u;
            }
        const uint64_t g = blockIdx.x;
        if (lane == 0)
            __hip_atomic_store(agg + g, agg_pack(v[0] | (v[1] << 1), h[0], h[1], c[0], c[1]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        uint32_t vin = 0, K = 0, vb = 0;
        entries_look_back(agg, pre, g, lane, vin, K, vb);
        const uint32_t vout = v[vin], k_out = K + h[vin], vb_out = vb + c[vin];
        if (lane == 0) {
            __hip_atomic_store(pre + g, pre_pack(vout, k_out, vb_out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gstate[0] = vin;
            gstate[1] = K;
            gstate[2] = vb;
            if (g + 1 == groups) {
                result->status = PQ_OK;
                result->n_entries = k_out;
                result->n_values = vb_out;
                result->reserved = 0;
                offsets[k_out] = vb_out;
            }
        }
    }
    __syncthreads();
    if (steps == 0u) return;

    // ---- the state that enters this tile, then its entries (as entries_write_kernel) ----
    uint32_t vin = uni(gstate[0]), K = uni(gstate[1]), vb = uni(gstate[2]);
    for (uint32_t w = 0; w < wave; ++w) {
        K += uni(wsum[w][1 + vin]);
        vb += uni(wsum[w][3 + vin]);
        vin = (uni(wsum[w][0]) >> vin) & 1u;
    }
    const uint64_t xb = asm_chain_in(bc, vin);
    uint64_t mx, my_v;
    uint32_t vo;
    asm_step(my_c, my_d, (uint32_t)(xb >> lane) & 1u, mx, my_v, vo);
    const uint64_t my_h = ~(my_c & mx) & my_valid;
    my_v &= my_valid;
    const uint32_t hcnt = (uint32_t)__builtin_popcountll(my_h), vcnt = (uint32_t)__builtin_popcountll(my_v);
    const uint32_t tile_entries = wave_sum_u32(hcnt);
    const uint32_t need = (uint32_t)__builtin_amdgcn_ballot_w64((my_h & ~my_v) != 0ull);

    const uint32_t n32 = (uint32_t)n;  // n < 2^30 on this path
    if (!repeated && (K & 63u) == 0u && steps == (uint32_t)kAsmSteps && base + kAsmTile <= n &&
        (uint64_t)K + kAsmTile <= n && __builtin_amdgcn_ballot_w64((dbits >> 8) != 0u) == 0ull) {
        // A whole tile of a column that is not REPEATED: every position opens an entry, so entry e is
        // position e, a step's validity bits are its value mask as it stands (lane s holds those of
        // step s and stores them itself) and nothing goes through LDS.
        // (a whole tile lies inside the column, so its 64 validity words lie inside the caller's array)
        if (lane < kAsmSteps) {
            validity[(K >> 5) + 2u * (uint32_t)lane] = (uint32_t)my_v;
            validity[(K >> 5) + 2u * (uint32_t)lane + 1u] = (uint32_t)(my_v >> 32);
        }
        uint32_t vbs = vb;
        uint32_t *op = offsets + K + lane;
        int32_t *ep = entry_def + K + lane;
#pragma unroll
        for (uint32_t s = 0; s < (uint32_t)kAsmSteps; ++s) {
            const uint64_t v = readlane_u64(my_v, s);
            const bool vbit = __builtin_amdgcn_inverse_ballot_w64(v);
            const uint32_t vr = __builtin_amdgcn_mbcnt_hi((uint32_t)(v >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)v, 0u));
            // written once, never read here: past the caches, so that they keep the levels still to come
            __builtin_nontemporal_store(vbs + vr, op + s * kWave);
            __builtin_nontemporal_store(vbit ? max_def : (int32_t)((dkeep[s >> 2] >> (8u * (s & 3u))) & 0xffu),
                                        ep + s * kWave);
            vbs += (uint32_t)__builtin_popcountll(v);
        }
        return;
    }

    uint32_t *vb32 = vbytes[wave];
    uint8_t *vb8 = reinterpret_cast<uint8_t *>(vb32);
#pragma unroll
    for (int k = 0; k < kAsmVbWords / kWave; ++k) vb32[lane + kWave * k] = 0;
    wave_fence();
    const uint32_t shift = K & 31u;
    // A step's two masks come to the scalar side (four v_readlane); whether a lane's position opens
    // an entry, and whether that entry has a value, is then the mask itself as the execution mask
    // or select mask.  The entries and values in front of each step come from one scan over the
    // tile's 32 steps (lane s: step s) and two more v_readlane a step.
    const uint32_t e_before = K + wave_scan_u32(hcnt) - hcnt, v_before = vb + wave_scan_u32(vcnt) - vcnt;
    int32_t level_of_values = max_def;
    asm volatile("" : "+v"(level_of_values));  // in a vector register once, not moved there per step
    // Entries never outnumber positions: the bound only keeps a damaged summary from turning into a
    // store outside the caller's arrays, and a tile whose entries all lie inside them (every tile
    // of an undamaged column) does without it.  n < 2^30 on this path: byte offsets fit 32 bits,
    // and the two stores share one (no 64-bit address arithmetic per entry).
    uint8_t *vb8e = vb8 + shift - K;  // entry e's validity byte (the sum is formed once, not per step)
    auto emit = [&](uint32_t s, int32_t dl, auto bounded) {
        const uint64_t H = readlane_u64(my_h, s), v = readlane_u64(my_v, s);
        const bool head = __builtin_amdgcn_inverse_ballot_w64(H);
        const bool vbit = __builtin_amdgcn_inverse_ballot_w64(v);
        const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(H >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)H, 0u));
        const uint32_t vr = __builtin_amdgcn_mbcnt_hi((uint32_t)(v >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)v, 0u));
        const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)e_before, (int)s) + r;
        if (head && (!decltype(bounded)::value || e < n32)) {
            const uint32_t at = e << 2;
            *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(offsets) + at) =
                (uint32_t)__builtin_amdgcn_readlane((int)v_before, (int)s) + vr;
            *reinterpret_cast<int32_t *>(reinterpret_cast<uint8_t *>(entry_def) + at) = vbit ? level_of_values : dl;
            vb8e[e] = vbit ? (uint8_t)1 : (uint8_t)0;
        }
    };
    if (__builtin_amdgcn_ballot_w64((dbits >> 8) != 0u) == 0ull && (uint64_t)K + tile_entries <= (uint64_t)n32) {
        // a whole tile whose levels were kept
#pragma unroll
        for (uint32_t s = 0; s < (uint32_t)kAsmSteps; ++s)
            emit(s, (int32_t)((dkeep[s >> 2] >> (8u * (s & 3u))) & 0xffu), std::false_type{});
    } else if (__builtin_amdgcn_ballot_w64((dbits >> 8) != 0u) == 0ull) {
#pragma unroll
        for (uint32_t s = 0; s < (uint32_t)kAsmSteps; ++s)
            emit(s, (int32_t)((dkeep[s >> 2] >> (8u * (s & 3u))) & 0xffu), std::true_type{});
    } else {
        for (uint32_t s0 = 0; s0 < steps; s0 += 8) {
            int32_t dls[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                dls[j] = max_def;
                const uint64_t pos = base + (uint64_t)(s0 + j) * kWave + lane;
                if (((need >> (s0 + j)) & 1u) != 0u && pos < n) dls[j] = def[pos];
            }
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                if (s0 + j >= steps) break;
                emit(s0 + j, dls[j], std::true_type{});
            }
        }
    }
    wave_fence();
    const uint32_t total_bits = shift + min(tile_entries, (uint32_t)kAsmTile);
    const uint32_t w0 = K >> 5;
    const uint64_t words_cap = (n + 31) / 32;
    const uint32_t nwords = w0 >= words_cap ? 0u : (uint32_t)min((uint64_t)((total_bits + 31u) >> 5), words_cap - w0);
    for (uint32_t wi = (uint32_t)lane; wi < nwords; wi += kWave) {
        uint32_t d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = vb32[8 * wi + k];
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) word |= (((d[k] * 0x01020408u) >> 24) & 0xfu) << (4 * k);
        const bool whole = (wi != 0u || shift == 0u) && 32u * wi + 32u <= total_bits;
        if (whole)
            validity[w0 + wi] = word;
        else if (word != 0u)
            atomicOr(validity + w0 + wi, word);
    }
}

// 32 levels -> W dwords per thread.
template <int W>
__global__ __launch_bounds__(256) void levels_pack_kernel(const int32_t *__restrict__ levels, uint64_t n,
                                                          uint8_t *__restrict__ out)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t first = t * 32;
    if (first >= n) return;
    uint32_t v[32];
    if (first + 32 <= n) {
        const uint4 *p = reinterpret_cast<const uint4 *>(levels + first);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint4 q = p[k];
            v[4 * k] = q.x;
            v[4 * k + 1] = q.y;
            v[4 * k + 2] = q.z;
            v[4 * k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = first + k < n ? (uint32_t)levels[first + k] : 0u;
    }
    uint32_t word[W];
    uint64_t acc = 0;
    int fill = 0, nw = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        acc |= (uint64_t)(v[k] & ((1u << W) - 1u)) << fill;
        fill += W;
        if (fill >= 32) {
            word[nw++] = (uint32_t)acc;
            acc >>= 32;
            fill -= 32;
        }
    }
    const uint64_t groups = (n + 7) / 8 - 4 * t;  // groups this thread writes (>= 1)
    if (groups >= 4) {
        uint32_t *o = reinterpret_cast<uint32_t *>(out) + t * W;
#pragma unroll
        for (int k = 0; k < W; ++k) o[k] = word[k];
    } else {
        uint8_t *o = out + t * 4 * W;
        const uint32_t bytes = (uint32_t)groups * W;
#pragma unroll
        for (int k = 0; k < W; ++k)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if ((uint32_t)(4 * k + b) < bytes) o[4 * k + b] = (uint8_t)(word[k] >> (8 * b));
    }
}

template <int W>
static void launch_pack_w(hipStream_t s, const int32_t *d_levels, uint64_t n, uint8_t *d_out)
{
    const uint64_t threads = (n + 31) / 32;
    hipLaunchKernelGGL(levels_pack_kernel<W>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, d_levels, n,
                       d_out);
}

hipError_t launch_levels_pack(hipStream_t s, const int32_t *d_levels, uint64_t n, int w, uint8_t *d_out)
{
    if (n == 0 || w == 0) return hipSuccess;
    switch (w) {
#define PQ_PACK_CASE(W) case W: launch_pack_w<W>(s, d_levels, n, d_out); break;
        PQ_PACK_CASE(1) PQ_PACK_CASE(2) PQ_PACK_CASE(3) PQ_PACK_CASE(4) PQ_PACK_CASE(5) PQ_PACK_CASE(6)
        PQ_PACK_CASE(7) PQ_PACK_CASE(8) PQ_PACK_CASE(9) PQ_PACK_CASE(10) PQ_PACK_CASE(11) PQ_PACK_CASE(12)
        PQ_PACK_CASE(13) PQ_PACK_CASE(14) PQ_PACK_CASE(15) PQ_PACK_CASE(16)
#undef PQ_PACK_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

size_t entries_workspace_bytes(uint64_t n)
{
    const uint64_t tiles = (n + kAsmTile - 1) / kAsmTile;
    const uint64_t groups = (tiles + kAsmWG / kWave - 1) / (kAsmWG / kWave);
    return (size_t)(tiles * kAsmSteps * 16 + tiles * 32 + groups * 32);
}

// Zeroes two ranges of 32-bit words in one launch.  A kernel rather than hipMemsetAsync: a caller
// may record these launches into a HIP graph, and memset nodes are not safe to replay there.
__global__ __launch_bounds__(256) void entries_clear_kernel(uint32_t *__restrict__ a, uint64_t na,
                                                            uint32_t *__restrict__ b, uint64_t nb)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < na; i += stride) a[i] = 0u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < nb; i += stride) b[i] = 0u;
}

static void launch_entries_clear(hipStream_t s, uint32_t *a, uint64_t na, uint32_t *b, uint64_t nb)
{
    const uint64_t most = na > nb ? na : nb;
    if (most == 0) return;
    const uint64_t blocks = (most + 1023) / 1024;  // four words a thread
    hipLaunchKernelGGL(entries_clear_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, a, na, b, nb);
}

hipError_t launch_column_entries(hipStream_t s, const int32_t *d_def, const int32_t *d_rep, uint64_t n,
                                 int32_t max_def, int32_t max_rep, int repeated, uint32_t *d_offsets,
                                 uint32_t *d_validity, int32_t *d_entry_def, pq_entries_result *d_result,
                                 void *d_work)
{
    const uint64_t tiles = (n + kAsmTile - 1) / kAsmTile;
    const uint64_t groups = (tiles + kAsmWG / kWave - 1) / (kAsmWG / kWave);
    uint64_t *masks = reinterpret_cast<uint64_t *>(d_work);
    uint32_t *tile_sum = reinterpret_cast<uint32_t *>(masks + tiles * kAsmSteps * 2);
    uint32_t *group_sum = tile_sum + tiles * 8;
    const uint64_t validity_words = (n + 31) / 32;
    if (groups != 0 && n < kOnePassMaxN) {
        // one launch: the group records (two 64-bit words per group) start at zero.  Groups of
        // kOnePassWG / 64 tiles.  (Groups of 16 tiles instead of 4, to have fewer records to look
        // back over, were no faster: the time is in the vector instructions of the write phase.)
        const uint64_t big = (tiles + kOnePassWG / kWave - 1) / (kOnePassWG / kWave);
        uint64_t *agg = reinterpret_cast<uint64_t *>(d_work);
        launch_entries_clear(s, d_validity, validity_words, reinterpret_cast<uint32_t *>(agg), big * 4);
        hipLaunchKernelGGL(entries_onepass_kernel<kOnePassWG>, dim3((unsigned)big), dim3(kOnePassWG), 0, s, d_def, d_rep,
                           n, max_def, max_rep, repeated, agg, agg + big, big, d_offsets, d_validity, d_entry_def, d_result);
        return hipGetLastError();
    }
    launch_entries_clear(s, d_validity, validity_words, nullptr, 0);
    if (groups != 0)
        hipLaunchKernelGGL(entries_summary_kernel, dim3((unsigned)groups), dim3(kAsmWG), 0, s, d_def, d_rep, n, max_def,
                           max_rep, repeated, masks, tile_sum, group_sum);
    hipLaunchKernelGGL(entries_scan_kernel, dim3(1), dim3(kAsmScanWG), 0, s, group_sum, groups, d_offsets, d_result);
    if (groups != 0)
        hipLaunchKernelGGL(entries_write_kernel, dim3((unsigned)groups), dim3(kAsmWG), 0, s, d_def, n, max_def, masks,
                           tile_sum, group_sum, d_offsets, d_validity, d_entry_def);
    return hipGetLastError();
}

// ---- column entries without any allocation ---------------------------------------------------
// The same launches over a workspace that is part of the library's device image: a file-scope
// device array, one copy per GPU, held until the process ends.  In host code the array's name is
// only a stand-in for the symbol; the device address comes from hipGetSymbolAddress, asked once
// per device (static_scratch_init, called when a context is created on the device) and kept in
// g_dev[device].
//
// Every context and stream of a GPU shares the one array, and the one-pass kernel's workgroups wait
// on the records in it: a second call that cleared them under a running first call would leave the
// first one's workgroups waiting for good.  So the calls of a GPU are put in order here: each call
// makes its stream wait for an event recorded behind the call before it (on whatever stream that
// was) and records the event again behind itself, all under the device's mutex.  Waiting on and
// recording an event allocate nothing.  A stream that is being captured into a graph is left
// alone (an event recorded outside a capture cannot be waited for inside one): the launches it
// records are ordered by the graph, and keeping a replay apart from other calls on the GPU is the
// caller's business.
constexpr size_t kStaticScratchBytes = 8u << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
struct ScratchDevice {
    std::mutex m;
    void *ptr = nullptr;             // device address of `scratch` on this GPU (null: not asked yet)
    hipError_t err = hipSuccess;     // why it is null after asking
    hipEvent_t last = nullptr;       // recorded behind the most recent call
    hipStream_t last_stream = nullptr;
    bool recorded = false;
};
static ScratchDevice g_dev[kMaxDevices];

size_t static_scratch_bytes() { return kStaticScratchBytes; }

// Once per device, with that device current.
hipError_t static_scratch_init(int device)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    ScratchDevice &D = g_dev[device];
    std::lock_guard<std::mutex> lock(D.m);
    if (D.ptr) return hipSuccess;
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e == hipSuccess && !D.last) e = hipEventCreateWithFlags(&D.last, hipEventDisableTiming);
    D.err = e;
    if (e != hipSuccess) return e;
    D.ptr = ptr;
    return hipSuccess;
}

hipError_t launch_column_entries_static(int device, hipStream_t s, const int32_t *d_def, const int32_t *d_rep,
                                        uint64_t n, int32_t max_def, int32_t max_rep, int repeated,
                                        uint32_t *d_offsets, uint32_t *d_validity, int32_t *d_entry_def,
                                        pq_entries_result *d_result)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (entries_workspace_bytes(n) > kStaticScratchBytes) return hipErrorInvalidValue;
    ScratchDevice &D = g_dev[device];
    std::lock_guard<std::mutex> lock(D.m);
    if (!D.ptr) return D.err != hipSuccess ? D.err : hipErrorNotInitialized;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    hipError_t e = hipSuccess;
    if (!capturing && D.recorded && D.last_stream != s) e = hipStreamWaitEvent(s, D.last, 0);
    if (e != hipSuccess) return e;
    // the launcher clears the words of the workspace that have to start at zero
    e = launch_column_entries(s, d_def, d_rep, n, max_def, max_rep, repeated, d_offsets, d_validity, d_entry_def,
                              d_result, D.ptr);
    if (e != hipSuccess || capturing) return e;
    e = hipEventRecord(D.last, s);
    if (e == hipSuccess) {
        D.recorded = true;
        D.last_stream = s;
    }
    return e;
}

}  // namespace pq
