// spmm_swept.hip - the opt-in L2-swept SpMM: host plan + persistent kernel (see the section comment).
#include "spmm_device.h"

void free_swept(ngcf_csr *c)
{
    ngcf_csr::Swept &w = c->swept;
    if (w.own_ptr) (void)hipFree(w.own_ptr);
    if (w.own_blk) (void)hipFree(w.own_blk);
    if (w.barrier) (void)hipFree(w.barrier);
    if (w.e_col) (void)hipFree(w.e_col);
    if (w.e_val) (void)hipFree(w.e_val);
    if (w.e_row) (void)hipFree(w.e_row);
    if (w.own_dst) (void)hipFree(w.own_dst);
    if (w.heavy_row) (void)hipFree(w.heavy_row);
    if (w.heavy_seg_ptr) (void)hipFree(w.heavy_seg_ptr);
    w = ngcf_csr::Swept();
}


// ---------------------------------------------------------------------------------------------
// Swept SpMM: the L2-blocked form for large matrices (same product LE = L.E, NGCF.py:130).
//
// Why: on a graph without locality the row-wise kernel above misses L2 on ~9 of 10 gathered rows and runs at
// the chip's L2-miss rate (~8 TB/s of gathered bytes); rows served from L2 arrive 2-3x faster (measured with a
// synchronised sliding window, profiles/r01_window_lab.txt).  Here the gathered table is swept in column blocks
// that fit an XCD's 4 MiB L2 while every CU works on the same block, so a table row is fetched from memory
// once per XCD and re-used from L2 by the other output rows of that XCD that need it.
//
// How: a persistent grid of 256 workgroups (one per CU, 512 threads, 128 KiB of LDS).  Output rows are handed to
// "owners"; an owner is a quarter-wave (16 lanes x 16 B = one 64-float slice of a row) that keeps up to 16
// accumulator rows in LDS and walks its own edge list, which the host plan has sorted by (column block, row)
// and balanced so that every owner has about the same work in every block.  The walk is software-pipelined:
// list entries are fetched two 16-entry chunks ahead, the 16 gathers of the next chunk are issued before the
// current chunk is accumulated.  After each column block the workgroups of one XCD (HW_REG_XCC_ID) meet at a
// counter barrier whose spin is bounded: the barrier only keeps the sweep together for speed, correctness never
// depends on it (an owner touches nothing but its own LDS rows and adds in list order: deterministic result).
// Rows longer than the per-owner budget are dealt round-robin to several pieces whose partial sums are combined
// by spmm_fixup_kernel in a fixed order.  A slice is 64 floats, so d must be a multiple of 64 (other widths use
// the row-wise kernel); slices and owner rounds are walked one after the other inside the kernel.
// Status: opt-in (ngcf_csr_set_mode(csr, 2)).  On the C3 item rows it reaches 3.2 ms against 3.5-3.7 ms for the
// row-wise kernel (69 % L2 hits), still far from the 1.3 ms of a perfectly synchronised sweep: with 8 waves per
// CU the per-entry run/flush logic and the barrier imbalance dominate.
// ---------------------------------------------------------------------------------------------
static const int kSweptRPO = 16;                 // accumulator rows per owner
static const int kSweptOwnersPerWG = 32;         // 8 waves x 4 quarter-waves
static const int kSweptWGs = 256;                // one 512-thread workgroup per CU (128 KiB of LDS)
static const int kSweptGroups = 8;               // XCDs
static const int64_t kSweptUnused = INT64_MIN;

static int32_t swept_block_cols()
{
    // columns per block: block bytes / (64 floats * 4 B); default 2 MiB of table slice per block
    const char *e = getenv("NGCF_SWEPT_BLOCK_KB");
    int64_t kb = e ? atoll(e) : 2048;
    if (kb < 16) kb = 16;
    return (int32_t)std::max<int64_t>(kb * 1024 / 256, 64);
}

template <typename F>
static void parallel_for(int64_t n, F &&fn)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 1) nt = 1;
    if (n < 64 || nt == 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> th;
    const int64_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const int64_t lo = t * chunk, hi = std::min<int64_t>(n, lo + chunk);
        if (lo >= hi) break;
        th.emplace_back([&fn, lo, hi]() { fn(lo, hi); });
    }
    for (auto &t : th) t.join();
}

int build_swept_plan(ngcf_csr *c, hipStream_t stream)
{
    free_swept(c);
    ngcf_csr::Swept &w = c->swept;
    const int64_t n_rows = c->n_rows, nnz = c->nnz;
    if (n_rows == 0) return NGCF_OK;
    std::vector<int64_t> rp((size_t)n_rows + 1);
    std::vector<int32_t> col((size_t)std::max<int64_t>(nnz, 1));
    std::vector<float> val((size_t)std::max<int64_t>(nnz, 1));
    HIP_TRY(hipMemcpyAsync(rp.data(), c->rowptr, sizeof(int64_t) * rp.size(), hipMemcpyDeviceToHost, stream));
    if (nnz > 0) {
        HIP_TRY(hipMemcpyAsync(col.data(), c->colidx, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(val.data(), c->vals, sizeof(float) * (size_t)nnz, hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));

    const int64_t per_round = (int64_t)kSweptOwnersPerWG * kSweptWGs;       // 8192 owners are resident at a time
    const int64_t rounds = std::max<int64_t>(1, (n_rows + per_round * 13 - 1) / (per_round * 13));
    const int64_t target = per_round * rounds;
    int64_t T = std::max<int64_t>(64, (nnz + target - 1) / target);
    w.block_cols = swept_block_cols();

    // entries begin+off, begin+off+step, ... < end.  A long row is dealt out to its pieces round-robin so that every
    // piece covers the whole column range evenly (a contiguous cut would pile one piece's work into a few blocks)
    struct Piece { int64_t begin, end, dst, off, step; int64_t count() const { return end - begin <= off ? 0 : (end - begin - off + step - 1) / step; } };
    std::vector<Piece> pieces;
    std::vector<int32_t> heavy_row;
    std::vector<int64_t> heavy_ptr, own_first;
    int64_t n_partial = 0;
    for (int attempt = 0; attempt < 40; ++attempt) {
        // 1) pieces: a row, or a <=T-entry cut of a long row (partial sums, combined in piece order)
        pieces.clear();
        heavy_row.clear();
        heavy_ptr.assign(1, 0);
        n_partial = 0;
        for (int64_t r = 0; r < n_rows; ++r) {
            const int64_t b = rp[r], e = rp[r + 1], len = e - b;
            if (len <= T) {
                pieces.push_back({b, e, r, 0, 1});
            } else {
                const int64_t k = (len + T - 1) / T;
                heavy_row.push_back((int32_t)r);
                for (int64_t j = 0; j < k; ++j) pieces.push_back({b, e, -1 - n_partial++, j, k});
                heavy_ptr.push_back(n_partial);
            }
        }
        // 2) owners: consecutive pieces until the entry budget or kSweptRPO rows are reached
        own_first.assign(1, 0);
        int64_t edges = 0;
        int rows_in = 0;
        for (size_t i = 0; i < pieces.size(); ++i) {
            const int64_t len = pieces[i].count();
            if (rows_in == kSweptRPO || (rows_in > 0 && edges + len > T)) {
                own_first.push_back((int64_t)i);
                edges = 0;
                rows_in = 0;
            }
            edges += len;
            ++rows_in;
        }
        own_first.push_back((int64_t)pieces.size());
        if ((int64_t)own_first.size() - 1 <= target) break;
        T += std::max<int64_t>(1, T / 16);          // too many owners for the resident grid: raise the budget
    }
    const int64_t n_owners = (int64_t)own_first.size() - 1;
    const int64_t n_owners_pad = align_up(n_owners, per_round);
    int32_t col_lo = INT32_MAX, col_hi = 0;
    for (int64_t x = 0; x < nnz; ++x) {
        col_lo = std::min(col_lo, col[x]);
        col_hi = std::max(col_hi, col[x]);
    }
    if (nnz == 0) col_lo = 0;
    w.col_lo = col_lo;
    const int64_t n_blocks = std::max<int64_t>(1, ((int64_t)col_hi - col_lo + w.block_cols) / w.block_cols);
    std::vector<int64_t> own_ptr((size_t)n_owners_pad + 1, 0);
    std::vector<int64_t> own_dst((size_t)n_owners_pad * kSweptRPO, kSweptUnused);
    for (int64_t o = 0; o < n_owners; ++o) {
        int64_t cnt = 0;
        for (int64_t i = own_first[o]; i < own_first[o + 1]; ++i) {
            cnt += pieces[i].count();
            own_dst[(size_t)(o * kSweptRPO + (i - own_first[o]))] = pieces[i].dst;
        }
        if (cnt >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "swept plan: owner list too long");
        own_ptr[(size_t)o + 1] = own_ptr[(size_t)o] + cnt;
    }
    for (int64_t o = n_owners; o < n_owners_pad; ++o) own_ptr[(size_t)o + 1] = own_ptr[(size_t)o];
    // 3) per owner: stable counting sort of its entries by (column block, local row); block offsets kept
    std::vector<int32_t> e_col((size_t)std::max<int64_t>(nnz, 1));
    std::vector<float> e_val((size_t)std::max<int64_t>(nnz, 1));
    std::vector<uint8_t> e_row((size_t)std::max<int64_t>(nnz, 1));
    const int32_t bc = w.block_cols;
    std::vector<int32_t> own_blk((size_t)n_owners_pad * (size_t)n_blocks, 0);
    parallel_for(n_owners, [&](int64_t lo, int64_t hi) {
        std::vector<int64_t> hist((size_t)n_blocks * kSweptRPO + 1);
        for (int64_t o = lo; o < hi; ++o) {
            std::fill(hist.begin(), hist.end(), 0);
            for (int64_t i = own_first[o]; i < own_first[o + 1]; ++i) {
                const int lr = (int)(i - own_first[o]);
                for (int64_t x = pieces[i].begin + pieces[i].off; x < pieces[i].end; x += pieces[i].step)
                    hist[(size_t)((col[x] - col_lo) / bc) * kSweptRPO + lr + 1]++;
            }
            for (size_t k = 1; k < hist.size(); ++k) hist[k] += hist[k - 1];
            for (int64_t bq = 0; bq < n_blocks; ++bq)      // where block bq ends inside this owner's list
                own_blk[(size_t)o * (size_t)n_blocks + (size_t)bq] = (int32_t)hist[(size_t)(bq + 1) * kSweptRPO];
            const int64_t base = own_ptr[(size_t)o];
            for (int64_t i = own_first[o]; i < own_first[o + 1]; ++i) {
                const int lr = (int)(i - own_first[o]);
                for (int64_t x = pieces[i].begin + pieces[i].off; x < pieces[i].end; x += pieces[i].step) {
                    const int64_t pos = base + hist[(size_t)((col[x] - col_lo) / bc) * kSweptRPO + lr]++;
                    e_col[(size_t)pos] = col[x];
                    e_val[(size_t)pos] = val[x];
                    e_row[(size_t)pos] = (uint8_t)lr;
                }
            }
        }
    });
    if (getenv("NGCF_SWEPT_DEBUG")) {
        int64_t mx = 0, nz = 0;
        for (int64_t o = 0; o < n_owners; ++o) {
            mx = std::max(mx, own_ptr[(size_t)o + 1] - own_ptr[(size_t)o]);
            nz += own_ptr[(size_t)o + 1] > own_ptr[(size_t)o];
        }
        fprintf(stderr, "[swept plan] rows %lld nnz %lld owners %lld (pad %lld, non-empty %lld) rounds %lld T %lld max/owner %lld "
                        "pieces %zu partial %lld blocks %lld x %d cols\n", (long long)n_rows, (long long)nnz, (long long)n_owners,
                (long long)n_owners_pad, (long long)nz, (long long)(n_owners_pad / per_round), (long long)T, (long long)mx,
                pieces.size(), (long long)n_partial, (long long)n_blocks, (int)bc);
    }
    // 4) upload
    w.n_owners = n_owners_pad;
    w.n_rounds = (int32_t)(n_owners_pad / per_round);
    w.n_blocks = (int32_t)n_blocks;
    w.n_entries = nnz;
    w.n_partial = n_partial;
    w.n_heavy = (int64_t)heavy_row.size();
    HIP_TRY(hipMalloc(&w.own_ptr, sizeof(int64_t) * own_ptr.size()));
    HIP_TRY(hipMalloc(&w.own_dst, sizeof(int64_t) * own_dst.size()));
    HIP_TRY(hipMalloc(&w.own_blk, sizeof(int32_t) * own_blk.size()));
    HIP_TRY(hipMalloc(&w.barrier, sizeof(uint32_t) * 32 * kSweptGroups));
    HIP_TRY(hipMemcpyAsync(w.own_blk, own_blk.data(), sizeof(int32_t) * own_blk.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMalloc(&w.e_col, sizeof(int32_t) * e_col.size()));
    HIP_TRY(hipMalloc(&w.e_val, sizeof(float) * e_val.size()));
    HIP_TRY(hipMalloc(&w.e_row, e_row.size()));
    HIP_TRY(hipMemcpyAsync(w.own_ptr, own_ptr.data(), sizeof(int64_t) * own_ptr.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(w.own_dst, own_dst.data(), sizeof(int64_t) * own_dst.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(w.e_col, e_col.data(), sizeof(int32_t) * e_col.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(w.e_val, e_val.data(), sizeof(float) * e_val.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(w.e_row, e_row.data(), e_row.size(), hipMemcpyHostToDevice, stream));
    if (w.n_heavy > 0) {
        HIP_TRY(hipMalloc(&w.heavy_row, sizeof(int32_t) * heavy_row.size()));
        HIP_TRY(hipMalloc(&w.heavy_seg_ptr, sizeof(int64_t) * heavy_ptr.size()));
        HIP_TRY(hipMemcpyAsync(w.heavy_row, heavy_row.data(), sizeof(int32_t) * heavy_row.size(), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(w.heavy_seg_ptr, heavy_ptr.data(), sizeof(int64_t) * heavy_ptr.size(), hipMemcpyHostToDevice, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    return NGCF_OK;
}

// value of lane U of this lane's 16-lane row (DPP row_newbcast: one VALU op, no LDS round trip)
template <int U> __device__ inline int row_bcast(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, 0x150 + U, 0xf, 0xf, false);
}
template <int U> __device__ inline float row_bcast(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + U, 0xf, 0xf, false));
}

// add a 16-lane x float4 accumulator into the owner's LDS row (plain read-modify-write: only this quarter-wave
// ever touches the row; LDS float atomics were measured slower and erratic here)
__device__ inline void swept_flush(float *__restrict__ rowp, float4 a)
{
    float4 *p = reinterpret_cast<float4 *>(rowp);
    float4 t = *p;
    t.x += a.x;
    t.y += a.y;
    t.z += a.z;
    t.w += a.w;
    *p = t;
}

#define NGCF_SWEPT_THREADS (kSweptOwnersPerWG * 16)

// entries k*16 .. k*16+15 of an owner's list: lane l keeps entry l (column, value, local row)
struct SweptEntries {
    int c, r, cnt;
    float v;
};

__device__ inline SweptEntries swept_load_entries(const int32_t *__restrict__ e_col, const float *__restrict__ e_val,
                                                  const uint8_t *__restrict__ e_row, int64_t pos, int64_t end, int l, int idle_col)
{
    SweptEntries e;
    const int64_t left = end - pos;
    e.cnt = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
    e.c = idle_col;
    e.r = 0;
    e.v = 0.f;
    if (l < e.cnt) {
        e.c = e_col[pos + l];
        e.v = e_val[pos + l];
        e.r = e_row[pos + l];
    }
    return e;
}

// issue the 16 gathers of a chunk; idle slots re-read the chunk's first row (an L2 hit), never column 0
__device__ inline void swept_issue(float4 (&x)[16], const SweptEntries &e, int l, const float *__restrict__ Es, int64_t ldE)
{
    const int c0 = row_bcast<0>(e.c);
    const int c = l < e.cnt ? e.c : c0;
#define NGCF_GATHER(u) x[u] = *reinterpret_cast<const float4 *>(Es + (int64_t)row_bcast<u>(c) * ldE);
    NGCF_GATHER(0) NGCF_GATHER(1) NGCF_GATHER(2) NGCF_GATHER(3) NGCF_GATHER(4) NGCF_GATHER(5) NGCF_GATHER(6) NGCF_GATHER(7)
    NGCF_GATHER(8) NGCF_GATHER(9) NGCF_GATHER(10) NGCF_GATHER(11) NGCF_GATHER(12) NGCF_GATHER(13) NGCF_GATHER(14) NGCF_GATHER(15)
#undef NGCF_GATHER
}

// consecutive entries of one row are summed in registers and added to the owner's LDS row when the row changes
__device__ inline void swept_accumulate(const float4 (&x)[16], const SweptEntries &e, float *__restrict__ myacc, int &cur, float4 &a)
{
#define NGCF_ACCUM(u)                                   \
    if (u < e.cnt) {                                    \
        const int rr = row_bcast<u>(e.r);               \
        if (rr != cur) {                                \
            swept_flush(myacc + cur * 64, a);           \
            a = vzero4();                               \
            cur = rr;                                   \
        }                                               \
        a = vfma(row_bcast<u>(e.v), x[u], a);           \
    }
    NGCF_ACCUM(0) NGCF_ACCUM(1) NGCF_ACCUM(2) NGCF_ACCUM(3) NGCF_ACCUM(4) NGCF_ACCUM(5) NGCF_ACCUM(6) NGCF_ACCUM(7)
    NGCF_ACCUM(8) NGCF_ACCUM(9) NGCF_ACCUM(10) NGCF_ACCUM(11) NGCF_ACCUM(12) NGCF_ACCUM(13) NGCF_ACCUM(14) NGCF_ACCUM(15)
#undef NGCF_ACCUM
}

// Meeting point of the workgroups of one XCD after a column block.  Bounded spin: a group that is not resident
// together only loses the L2 re-use; it never hangs and never changes the result.
__device__ inline void swept_group_sync(unsigned *ctr, unsigned target, int &max_spin)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < max_spin)
            __builtin_amdgcn_s_sleep(1);
        if (max_spin > 0 && spins >= max_spin) max_spin = 0;   // the group is not resident together: stop waiting from now on
    }
    __syncthreads();
}

__global__ __launch_bounds__(NGCF_SWEPT_THREADS) void spmm_swept_kernel(
    const int64_t *__restrict__ own_ptr, const int32_t *__restrict__ own_blk, const int32_t *__restrict__ e_col,
    const float *__restrict__ e_val, const uint8_t *__restrict__ e_row, const int64_t *__restrict__ own_dst, int n_rounds,
    int n_blocks, int n_slices, const float *__restrict__ E, int64_t ldE, float *__restrict__ out, int64_t ldo,
    float *__restrict__ partial, int dp, unsigned *bar, int max_spin)
{
    __shared__ float acc_lds[kSweptOwnersPerWG * kSweptRPO * 64 + 4];   // 128 KiB of accumulators (+ the XCD id)
    const int q = threadIdx.x >> 4;          // owner slot in the workgroup
    const int l = threadIdx.x & 15;          // lane in the quarter-wave
    float *myacc = acc_lds + q * (kSweptRPO * 64) + l * 4;
    if (threadIdx.x == 0)
        reinterpret_cast<unsigned *>(acc_lds)[kSweptOwnersPerWG * kSweptRPO * 64] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    __syncthreads();
    unsigned *ctr = bar + reinterpret_cast<unsigned *>(acc_lds)[kSweptOwnersPerWG * kSweptRPO * 64] * 32;   // HW_REG_XCC_ID
    const unsigned members = gridDim.x / kSweptGroups;
    unsigned seq = 0;
    for (int slice = 0; slice < n_slices; ++slice) {
        const float *Es = E + slice * 64 + l * 4;
        for (int round = 0; round < n_rounds; ++round) {
            const int64_t owner = ((int64_t)round * gridDim.x + blockIdx.x) * kSweptOwnersPerWG + q;
#pragma unroll
            for (int r = 0; r < kSweptRPO; ++r) *reinterpret_cast<float4 *>(myacc + r * 64) = vzero4();
            const int64_t beg = own_ptr[owner], end = own_ptr[owner + 1];
            const int32_t *blk = own_blk + owner * (int64_t)n_blocks;
            // chunks this WAVE walks: the longest of its four owners
            int my_chunks = (int)((end - beg + 15) >> 4);
            my_chunks = max(my_chunks, __shfl_xor(my_chunks, 16));
            my_chunks = max(my_chunks, __shfl_xor(my_chunks, 32));
            int cur = 0, b = 0;
            int bend = n_blocks > 0 ? blk[0] : 0x7fffffff;     // end of block b in this owner's list (kept one block ahead)
            float4 a = vzero4();
            int idle_col = end > beg ? e_col[beg] : 0;
            // software pipeline: entries two chunks ahead, gathers one chunk ahead of the accumulation
            float4 xA[16], xB[16];
            SweptEntries eA = swept_load_entries(e_col, e_val, e_row, beg, end, l, idle_col);
            SweptEntries eB = swept_load_entries(e_col, e_val, e_row, beg + 16, end, l, idle_col);
            swept_issue(xA, eA, l, Es, ldE);
            for (int k = 0; k < my_chunks; k += 2) {
                // ---- chunk k (set A): prefetch entries k+2, issue gathers k+1, accumulate k
                SweptEntries eC = swept_load_entries(e_col, e_val, e_row, beg + (int64_t)(k + 2) * 16, end, l, idle_col);
                swept_issue(xB, eB, l, Es, ldE);
                swept_accumulate(xA, eA, myacc, cur, a);
                // a column block is finished once every owner of the wave has walked past its end
                while (b < n_blocks) {
                    int done = bend <= (k + 1) * 16 ? 1 : 0;
                    done &= __shfl_xor(done, 16);
                    done &= __shfl_xor(done, 32);
                    if (!done) break;
                    swept_group_sync(ctr, members * (++seq), max_spin);
                    ++b;
                    bend = b < n_blocks ? blk[b] : 0x7fffffff;
                }
                // ---- chunk k+1 (set B)
                eA = swept_load_entries(e_col, e_val, e_row, beg + (int64_t)(k + 3) * 16, end, l, idle_col);
                swept_issue(xA, eC, l, Es, ldE);
                swept_accumulate(xB, eB, myacc, cur, a);
                while (b < n_blocks) {
                    int done = bend <= (k + 2) * 16 ? 1 : 0;
                    done &= __shfl_xor(done, 16);
                    done &= __shfl_xor(done, 32);
                    if (!done) break;
                    swept_group_sync(ctr, members * (++seq), max_spin);
                    ++b;
                    bend = b < n_blocks ? blk[b] : 0x7fffffff;
                }
                eB = eA;
                eA = eC;
                // rotate: next iteration accumulates chunk k+2 from xA (issued above from eC) with entries eA = eC,
                // and needs eB = entries k+3
            }
            for (; b < n_blocks; ++b) swept_group_sync(ctr, members * (++seq), max_spin);   // every wave meets n_blocks times
            swept_flush(myacc + cur * 64, a);
            // write the owner's rows (its own LDS rows only: no barrier needed)
#pragma unroll 1
            for (int r = 0; r < kSweptRPO; ++r) {
                const int64_t dst = own_dst[owner * kSweptRPO + r];
                if (dst == kSweptUnused) continue;
                float *p = dst >= 0 ? out + dst * ldo : partial + (-1 - dst) * (int64_t)dp;
                *reinterpret_cast<float4 *>(p + slice * 64 + l * 4) = *reinterpret_cast<const float4 *>(myacc + r * 64);
            }
        }
    }
}


int launch_swept(const ngcf_csr *c, const float *E, int64_t ldE, int d, float *out, int64_t ldo, float *partial, int dp,
                 hipStream_t stream)
{
    const ngcf_csr::Swept &w = c->swept;
    static const int max_spin = getenv("NGCF_SWEPT_SPIN") ? atoi(getenv("NGCF_SWEPT_SPIN")) : 400;
    HIP_TRY(hipMemsetAsync(w.barrier, 0, sizeof(uint32_t) * 32 * kSweptGroups, stream));
    prof_mark(stream, 0);
    spmm_swept_kernel<<<dim3(kSweptWGs), NGCF_SWEPT_THREADS, 0, stream>>>(
        w.own_ptr, w.own_blk, w.e_col, w.e_val, w.e_row, w.own_dst, w.n_rounds, w.n_blocks, d / 64, E, ldE, out, ldo, partial,
        dp, w.barrier, max_spin);
    LAUNCH_CHECK();
    prof_mark(stream, 1);
    if (w.n_heavy > 0) {
        spmm_fixup_kernel<4><<<dim3((unsigned)((w.n_heavy + 3) / 4)), 256, 0, stream>>>(w.heavy_row, w.heavy_seg_ptr, w.n_heavy,
                                                                                        partial, dp, d, out, ldo);
        LAUNCH_CHECK();
    }
    return NGCF_OK;
}
