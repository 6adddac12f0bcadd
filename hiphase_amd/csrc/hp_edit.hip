// hp_edit.hip — batched Levenshtein distance (reference src/sequence_alignment.rs:7-38), the inner call of
// Variant::closest_allele_clip (reference src/data_types/variants.rs:624-641) used by local re-alignment.
//
// One wavefront per pair. The reference fills the full grid row by row; here a row is computed 64 cells at a
// time: with t[j] = min(prev[j] + 1, prev[j-1] + (a[j-1] != b[i])) the in-row dependency
// row[j] = min(t[j], row[j-1] + 1) unrolls to row[j] = j + min_{k<=j}(t[k] - k), i.e. a wave prefix-min.
// The shorter sequence is laid along the row (the distance is symmetric). Rows live in LDS when they fit,
// else in HBM scratch. u32 cells (distances <= max(len) < 2^32). No MFMA: byte compares and integer min/add.
#include "hp_common.h"

#include <algorithm>
#include <numeric>
#include <vector>

namespace hp {

struct EdPairDev {
    uint64_t a_off, b_off;   // into the byte pool; a = the shorter sequence (along the row)
    uint32_t a_len, b_len;
};
struct EdBatchDev {
    const EdPairDev* pairs;
    const uint32_t* order;
    uint32_t n_items;
    const uint8_t* bytes;
    uint64_t* out;
    uint32_t* scratch;        // [slots][2 * row_stride] for rows that do not fit LDS
    uint64_t row_stride;
    uint32_t lds_row_cap;     // cells per row available in LDS
};

#define EDEV __device__ __forceinline__
extern __shared__ __attribute__((aligned(16))) unsigned char ed_smem[];

EDEV uint32_t ed_lane() { return __lane_id(); }

// inclusive prefix-min over the wave (signed)
EDEV int32_t wave_prefix_min(int32_t v) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int32_t o = __shfl_up(v, s);
        if ((int)ed_lane() >= s) v = min(v, o);
    }
    return v;
}

EDEV void solve_pair(const EdBatchDev& B, uint32_t id, uint32_t slot) {
    const EdPairDev pr = B.pairs[id];
    const uint8_t* a = B.bytes + pr.a_off;
    const uint8_t* b = B.bytes + pr.b_off;
    const uint32_t l1 = pr.a_len, l2 = pr.b_len;
    const uint32_t lane = ed_lane();
    const bool in_lds = (l1 + 1) <= B.lds_row_cap;
    uint32_t* row0 = in_lds ? reinterpret_cast<uint32_t*>(ed_smem) : B.scratch + (size_t)slot * 2 * B.row_stride;
    uint32_t* row1 = row0 + (in_lds ? B.lds_row_cap : B.row_stride);
    // prev_row = 0..l1 (sequence_alignment.rs:11)
    for (uint32_t j = lane; j <= l1; j += 64) row0[j] = j;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    uint32_t* prev = row0;
    uint32_t* cur = row1;
    for (uint32_t i = 0; i < l2; ++i) {
        const uint8_t c2 = b[i];
        int32_t carry = INT32_MAX;  // min over all earlier columns of (t[k] - k)
        for (uint32_t base = 0; base <= l1; base += 64) {
            const uint32_t j = base + lane;
            int32_t t = INT32_MAX;
            if (j <= l1) {
                if (j == 0) t = (int32_t)(i + 1);  // row[0] = i + 1 (sequence_alignment.rs:15)
                else {
                    const uint32_t up = prev[j] + 1;                          // skip a character in v2
                    const uint32_t dg = prev[j - 1] + (a[j - 1] == c2 ? 0u : 1u);  // diagonal
                    t = (int32_t)min(up, dg);
                }
                t -= (int32_t)j;
            }
            int32_t m = wave_prefix_min(t);
            m = min(m, carry);
            if (j <= l1) cur[j] = (uint32_t)(m + (int32_t)j);
            carry = __shfl(m, 63);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        uint32_t* tmp = prev; prev = cur; cur = tmp;
    }
    uint32_t res = 0;
    if (lane == 0) res = prev[l1];
    res = (uint32_t)__builtin_amdgcn_readfirstlane((int)res);
    if (lane == 0) B.out[id] = res;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
}

__global__ void __launch_bounds__(64) hp_edit_kernel(EdBatchDev B) {
    const uint32_t slot = blockIdx.x, G = gridDim.x;
    for (uint32_t r = 0;; ++r) {
        const uint32_t base = r * G;
        if (base >= B.n_items) break;
        const uint32_t i = base + ((r & 1u) ? (G - 1u - slot) : slot);
        if (i < B.n_items) solve_pair(B, B.order[i], slot);
    }
}

}  // namespace hp

using namespace hp;

extern "C" int hp_edit_distance_batch(const hp_ed_pair* pairs, size_t n, uint64_t* out, int device_id) {
    if (n == 0) return HP_OK;
    if (!pairs || !out) { set_error("null argument"); return HP_ERR_ARG; }
    if (n > 0x7FFFFFFFull) { set_error("too many pairs"); return HP_ERR_ARG; }
    std::vector<EdPairDev> dp(n);
    std::vector<uint8_t> bytes;
    uint32_t max_short = 0;
    {
        size_t total = 16;
        for (size_t i = 0; i < n; ++i) total += (size_t)pairs[i].a_len + pairs[i].b_len;
        bytes.reserve(total);
    }
    for (size_t i = 0; i < n; ++i) {
        const hp_ed_pair& p = pairs[i];
        if ((p.a_len && !p.a) || (p.b_len && !p.b)) { set_error("pair %zu: null sequence", i); return HP_ERR_ARG; }
        const bool swap = p.a_len > p.b_len;  // shorter one along the row; Levenshtein distance is symmetric
        const uint8_t* s = swap ? p.b : p.a; const uint32_t sl = swap ? p.b_len : p.a_len;
        const uint8_t* l = swap ? p.a : p.b; const uint32_t ll = swap ? p.a_len : p.b_len;
        dp[i].a_off = bytes.size(); bytes.insert(bytes.end(), s, s + sl);
        dp[i].b_off = bytes.size(); bytes.insert(bytes.end(), l, l + ll);
        dp[i].a_len = sl; dp[i].b_len = ll;
        max_short = std::max(max_short, sl);
    }
    bytes.resize(bytes.size() + 16, 0);
    if (device_id < 0) device_id = hp_default_device();
    if (hp_set_device(device_id) != hipSuccess) { set_error("hipSetDevice(%d) failed - no usable GPU; there is no CPU fallback", device_id); return HP_ERR_HIP; }
    const int n_cu = partition_cu_count(device_id);
    // largest tables first (LPT). The order only balances the work list, so a counting sort over 65536 size classes
    // (O(n)) does as well as an exact sort
    std::vector<uint32_t> order(n);
    {
        uint64_t max_cells = 1;
        for (size_t i = 0; i < n; ++i) max_cells = std::max(max_cells, (uint64_t)dp[i].a_len * dp[i].b_len);
        constexpr uint32_t NB = 65536;
        auto bucket = [&](size_t i) { return NB - 1 - (uint32_t)((unsigned __int128)((uint64_t)dp[i].a_len * dp[i].b_len) * (NB - 1) / max_cells); };
        std::vector<uint32_t> start(NB + 1, 0);
        for (size_t i = 0; i < n; ++i) start[bucket(i) + 1]++;
        for (uint32_t b = 0; b < NB; ++b) start[b + 1] += start[b];
        for (size_t i = 0; i < n; ++i) order[start[bucket(i)]++] = (uint32_t)i;
    }
    // rows in LDS up to 2 048 cells (2 rows x 2 048 x 4 B = 16 KiB per wavefront) - but no more than the batch's longest row needs:
    // beside a resident graph-WFA launch set a compute unit has ~19 KB of LDS left, and the fallbacks of a block set (alleles of a
    // few hundred bases at most) then run four or more workgroups per compute unit instead of one
    const uint32_t lds_row_cap = std::min<uint32_t>(2048u, std::max<uint32_t>(64u, (max_short + 1u + 63u) & ~63u));
    const uint64_t row_stride = ((uint64_t)max_short + 1 + 63) & ~63ull;
    const uint32_t slots = (uint32_t)std::min<size_t>(n, (size_t)n_cu * 8);
    DevBuf d_pairs, d_order, d_bytes, d_out, d_scratch;
    int rc;
    if ((rc = d_pairs.alloc(n * sizeof(EdPairDev))) || (rc = d_order.alloc(n * 4)) || (rc = d_bytes.alloc(bytes.size())) ||
        (rc = d_out.alloc(n * 8)))
        return rc;
    if (max_short + 1 > lds_row_cap && (rc = d_scratch.alloc((size_t)slots * 2 * row_stride * 4)) != HP_OK) return rc;
    // the calling thread's own stream; the call waits for that stream only (a block stream's other stages keep running)
    hipStream_t stm = thread_stream(device_id);
    if (!stm) { set_error("stream creation failed"); return HP_ERR_HIP; }
    struct StreamDrain { hipStream_t s; ~StreamDrain() { (void)hipStreamSynchronize(s); } } drain{stm};
    // (dev_put / dev_get, hp_common.h: not the runtime's copies)
    struct IoDrain { hipStream_t s; bool armed; ~IoDrain() { if (armed) dev_io_abort(s); } } io{stm, true};
    if ((rc = dev_put(d_pairs.p, dp.data(), n * sizeof(EdPairDev), stm)) || (rc = dev_put(d_order.p, order.data(), n * 4, stm)) ||
        (rc = dev_put(d_bytes.p, bytes.data(), bytes.size(), stm)))
        return rc;
    EdBatchDev B{};
    B.pairs = d_pairs.as<EdPairDev>(); B.order = d_order.as<uint32_t>(); B.n_items = (uint32_t)n;
    B.bytes = d_bytes.as<uint8_t>(); B.out = d_out.as<uint64_t>(); B.scratch = d_scratch.as<uint32_t>();
    B.row_stride = row_stride; B.lds_row_cap = lds_row_cap;
    hipEvent_t e0, e1;
    HP_HIP_CHECK(hipEventCreate(&e0));
    HP_HIP_CHECK(hipEventCreate(&e1));
    HP_HIP_CHECK(hipEventRecord(e0, stm));
    hipLaunchKernelGGL(hp_edit_kernel, dim3(slots), dim3(64), (size_t)lds_row_cap * 2 * 4, stm, B);
    HP_HIP_CHECK(hipGetLastError());
    HP_HIP_CHECK(hipEventRecord(e1, stm));
    if ((rc = dev_get(out, d_out.p, n * 8, stm)) != HP_OK) return rc;
    io.armed = false;
    if ((rc = dev_io_sync(stm)) != HP_OK) return rc;
    float kms = 0.f;
    HP_HIP_CHECK(hipEventElapsedTime(&kms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    g_last_kernel_ms = kms;
    return HP_OK;
}
