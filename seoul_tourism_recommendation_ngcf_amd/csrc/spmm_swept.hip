// spmm_swept.hip - the L2-swept SpMM for long-lived matrices: host plan + persistent kernel.
//
// Same product LE = L.E (NGCF.py:130).  Why: on a graph without locality the row-wise kernels miss L2 on most
// gathered rows and run at the chip's L2-miss rate (~8 TB/s of gathered bytes); rows served from L2 arrive 2-3x
// faster (profiles/r01_window_lab.txt).  Here the gathered table is swept in column windows of a few MiB while
// every CU works on the same window, so a table row is fetched from memory once per XCD and re-used from that
// XCD's L2 by the other output rows that need it.
//
// How: a persistent grid of 256 workgroups (one per CU).  Every wave owns up to RW output rows ("wave task") whose
// accumulators - one 64-float slice each - stay in LDS for a whole sweep, so the chip's LDS holds 147 K row slices
// at once; d is walked slice by slice, the rows in "row passes" when they do not fit at once.  The host plan deals
// rows (long rows in strided pieces) to the wave tasks so that all tasks carry the same number of entries, and
// lays each task's entries out window by window (8 MiB of table slice), inside a window in ascending column order:
// waves that enter a window together walk the table left to right together.  A round = one wave instruction = four
// entries (16 lanes x 16 B each).  The contribution of an entry is added to its LDS row with a plain
// read-modify-write (LDS float atomics were measured 15x slower), so the four entries of a round must belong to four
// different rows: the rounds are filled greedily in column order, an entry whose row is already in the round waits
// for the next one; slots that stay empty (~1 % on C3) point at a spare LDS row.  A wave touches nothing but its own
// LDS rows and adds in list order: the result is deterministic.
// The sweep is kept together per XCD (HW_REG_XCC_ID) by counters: a wave may enter window s once every workgroup
// of its XCD has left window s-1-lead behind.  The spin is bounded and only serves speed - a grid that is not
// resident together loses the L2 re-use, it never hangs and never changes the result.
// Rows cut into pieces leave partial sums in the workspace; spmm_fixup_kernel adds them in piece order.
// Measured on C3 (1x MI355X, d=128): item rows 1.83-1.93 ms vs 3.5 ms row-wise, user rows 1.84-1.94 vs 2.46 ms d-sliced.
#include "spmm_device.h"

namespace {
// Geometry of a part: LPE lanes share one entry, i.e. a slice of SW = 4 LPE floats of the gathered row per wave instruction;
// EPR = 64 / LPE entries per round, 16 rounds per chunk.  LPE = 16: 64-float (256-byte) slices, four entries per round, 576
// accumulator rows per workgroup.  LPE = 32 (r03): 128-float (512-byte) slices - at d = 128 the whole row in ONE sweep, two entries
// per round, 288 accumulator rows per workgroup: twice the row passes, half the slices, every stored entry read once instead of
// twice, four consecutive 128-byte lines per gather instead of two.
constexpr int kLdsBytes = 36 * 16 * 256;  // accumulator bytes per workgroup (144 KiB of LDS, + one spare row per wave)
constexpr int kSweptWGs = 256;            // one workgroup per CU
constexpr int kRing = 8;                  // rotating sweep counters per XCD
constexpr int kRowBits = 24;              // e_pack: column in bits 0..23, local row in bits 24..30
constexpr int kColMask = (1 << kRowBits) - 1;

template <typename F>
void parallel_for(int64_t n, F &&fn, int64_t min_parallel = 64)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 1) nt = 1;
    if ((int64_t)nt > n) nt = (unsigned)std::max<int64_t>(n, 1);
    if (n < min_parallel || nt == 1) {
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

void free_part(ngcf_csr::Swept::Part &p)
{
    if (p.tptr) (void)hipFree(p.tptr);
    if (p.e_pack) (void)hipFree(p.e_pack);
    if (p.e_val) (void)hipFree(p.e_val);
    if (p.dst) (void)hipFree(p.dst);
    if (p.prow) (void)hipFree(p.prow);
    if (p.heavy_row) (void)hipFree(p.heavy_row);
    if (p.heavy_seg_ptr) (void)hipFree(p.heavy_seg_ptr);
    p = ngcf_csr::Swept::Part();
}

void free_segset(ngcf_csr::SegSet &s)
{
    if (s.seg_row) (void)hipFree(s.seg_row);
    if (s.seg_begin) (void)hipFree(s.seg_begin);
    if (s.heavy_row) (void)hipFree(s.heavy_row);
    if (s.heavy_seg_ptr) (void)hipFree(s.heavy_seg_ptr);
    s = ngcf_csr::SegSet();
}

template <typename T>
int upload(T **dptr, const std::vector<T> &h, hipStream_t stream)
{
    HIP_TRY(hipMalloc(dptr, sizeof(T) * std::max<size_t>(h.size(), 1)));
    if (!h.empty()) HIP_TRY(hipMemcpyAsync(*dptr, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, stream));
    return NGCF_OK;
}

// entries begin+off, begin+off+step, ... < end of one row.  A long row is dealt out to its pieces round-robin so
// that every piece covers the whole column range evenly (a contiguous cut would pile a piece's work into a few windows)
struct Piece {
    int64_t row;            // >= 0 row of the part, < 0 partial row -1-p of the part
    int64_t begin, end, count;
    int32_t off, step;
    int64_t src;            // the row of the part its entries come from (== row when the row is not cut)
};

// Plan of the rows [row_lo, row_hi).  Leaves part.waves == 0 when the part is not worth it (mode 3) or the shape does
// not fit the entry encoding.
int build_part(const ngcf_csr *c, const ngcf_csr::RowGroup &grp, bool force, hipStream_t stream, ngcf_csr::Swept::Part &part, int LPE)
{
    const int kSW = LPE * 4, kEPR = 64 / LPE, kLdsRows = kLdsBytes / (kSW * 4);        // the part's geometry (see the top of the file)
    const int64_t row_lo = grp.begin, row_hi = grp.end, n = row_hi - row_lo;
    const int32_t col_lo = grp.col_lo, col_hi = grp.col_hi;
    if (n <= 0 || col_hi < col_lo) return NGCF_OK;
    if (col_hi > kColMask) return NGCF_OK;                           // column does not fit the packed entry
    std::vector<int64_t> rp((size_t)n + 1);
    HIP_TRY(hipMemcpyAsync(rp.data(), c->rowptr + row_lo, sizeof(int64_t) * rp.size(), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    const int64_t e0 = rp[0], nnz = rp[(size_t)n] - e0;
    if (nnz <= 0) return NGCF_OK;
    if (!force && nnz < ((int64_t)1 << 22)) return NGCF_OK;          // small products are launch-bound, not L2-bound
    // workgroup shape: 16 waves x 36 rows.  (Round 1 ran the parts that need three or more row passes as 8 x 72: fewer,
    // longer wave tasks fetched 30 % less.  With the lag-driven wave priorities of the kernel 16 waves are faster there
    // too - C3 user rows 1.60 vs 1.93 ms, profiles/r02_swept_lab.txt.)  Lab builds (-DNGCF_LAB) keep the 8-wave shape behind the option swept_waves.
    const int64_t cap = (int64_t)kSweptWGs * kLdsRows;               // output rows resident in LDS at a time
    int waves = 16;
#ifdef NGCF_LAB
    if (ngcf_opts().swept_waves == 8) waves = 8;                      // the 8 x 72 shape is a lab instantiation
#endif
    const int RW = kLdsRows / waves;
    const int64_t n_wave_slots = (int64_t)kSweptWGs * waves;
    if (!force) {
        // expected uses of a fetched table row inside one XCD during one sweep: its column degree times the share
        // of the output rows that the XCD holds in LDS.  Below ~3 the sweep costs more than the misses it saves;
        // a table slice that fits the L2s anyway needs no sweep either.
        const double span = (double)col_hi - col_lo + 1;
        const double reuse = (double)nnz / span * std::min(1.0, (double)(cap / 8) / (double)n);
        if (reuse < 3.0 || span * kSW * 4 < (double)(8 << 20)) return NGCF_OK;
    }
    std::vector<int32_t> col((size_t)nnz);    // only now the entries come to the host
    std::vector<float> val((size_t)nnz);
    HIP_TRY(hipMemcpyAsync(col.data(), c->colidx + e0, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(val.data(), c->vals + e0, sizeof(float) * (size_t)nnz, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (auto &x : rp) x -= e0;
    // 1) pieces and wave tasks: all tasks get the same number of entries and at most RW rows
    int64_t n_rowpass = std::max<int64_t>(1, (n + cap - 1) / cap), n_tasks = 0, T = 0, n_partial = 0;
    std::vector<Piece> pieces;
    std::vector<int32_t> heavy_row;
    std::vector<int64_t> heavy_ptr;
    for (;; ++n_rowpass) {
        n_tasks = n_rowpass * n_wave_slots;
        T = std::max<int64_t>(64, (nnz + n_tasks - 1) / n_tasks);
        const int64_t Tp = std::max<int64_t>(64, T / std::max(2, ngcf_opts().swept_cut));   // rows are cut well below a task's share
        pieces.clear();
        heavy_row.clear();
        heavy_ptr.assign(1, 0);
        n_partial = 0;
        for (int64_t r = 0; r < n; ++r) {
            const int64_t b = rp[(size_t)r], e = rp[(size_t)r + 1], len = e - b;
            if (len <= Tp) {
                pieces.push_back({r, b, e, len, 0, 1, r});
            } else {
                const int64_t k = (len + Tp - 1) / Tp;
                heavy_row.push_back((int32_t)(row_lo + r));
                for (int64_t j = 0; j < k; ++j) pieces.push_back({-1 - n_partial++, b, e, (len - j + k - 1) / k, (int32_t)j, (int32_t)k, r});
                heavy_ptr.push_back(n_partial);
            }
        }
        if ((int64_t)pieces.size() <= n_tasks * RW) break;
        if (n_rowpass > (int64_t)1 << 20) return fail(NGCF_ERR_ARG, "swept plan: cannot place %zu pieces", pieces.size());
    }
    if (n_tasks * RW >= (int64_t)1 << 31 || n_partial >= ((int64_t)1 << 31) - 2) return fail(NGCF_ERR_ARG, "swept plan: matrix too large");
    // Dealing.  Level by level, heaviest piece -> lightest task (every task gets at most one piece per level): all tasks end
    // within +-2 % of the same entry count.  That alone leaves WHERE in the table a task's entries lie to chance, and the
    // sweep pays for it: a wave whose rows happen to hold few entries in the first half of the table runs ahead of the
    // others by (missing entries) / (entries per column) - on the C3 item rows sigma = 1.1 MiB of table slice, 4 096 waves
    // spread over ~10 MiB against a 4 MiB L2 (measured with tools/swept_trace_lab.py, profiles/r02_swept_trace.txt; a table
    // row is then fetched again for the late waves).  So inside buckets of kDealBucket neighbouring (task, piece) pairs of a
    // level - equal loads to within a fraction of a percent - the pairing is chosen to cancel the low-order cosine moments
    // of every task's entry positions (position = share of all entries left of the column, so that a uniform sweep is the
    // target whatever the column popularity): m_k = sum over entries of cos(k pi x), k = 1..kDealK, are the sine-series
    // coefficients of the task's cumulative-count deviation; the greedy gives each piece to the task of its bucket whose
    // moments it cancels best.  Simulated on the C3 item rows: spread (p5..p95) 2.9 -> 0.9 MiB, worst wave 4.6 -> 2.0 MiB.
    constexpr int kDealK = 8, kDealBucket = 256;
    const bool balance_moments = !ngcf_opts().swept_no_moments;
    std::vector<float> mom;                                          // [piece][kDealK]
    if (balance_moments) {
        const int64_t span = (int64_t)col_hi - col_lo + 1;
        std::vector<int64_t> ccnt((size_t)span + 1, 0);
        for (int64_t x = 0; x < nnz; ++x) ccnt[(size_t)(col[(size_t)x] - col_lo) + 1]++;
        for (int64_t c = 0; c < span; ++c) ccnt[(size_t)c + 1] += ccnt[(size_t)c];
        std::vector<float> ctab((size_t)span * kDealK);
        parallel_for(span, [&](int64_t lo, int64_t hi) {
            for (int64_t c = lo; c < hi; ++c) {
                const double xpos = (0.5 * (double)(ccnt[(size_t)c] + ccnt[(size_t)c + 1])) / (double)nnz;
                for (int k = 0; k < kDealK; ++k) ctab[(size_t)c * kDealK + k] = (float)cos((k + 1) * 3.14159265358979323846 * xpos);
            }
        });
        mom.assign(pieces.size() * kDealK, 0.f);
        parallel_for((int64_t)pieces.size(), [&](int64_t lo, int64_t hi) {
            for (int64_t pi = lo; pi < hi; ++pi) {
                const Piece &pc = pieces[(size_t)pi];
                float acc[kDealK] = {0};
                for (int64_t x = pc.begin + pc.off; x < pc.end; x += pc.step) {
                    const float *t = &ctab[(size_t)(col[(size_t)x] - col_lo) * kDealK];
                    for (int k = 0; k < kDealK; ++k) acc[k] += t[k];
                }
                for (int k = 0; k < kDealK; ++k) mom[(size_t)pi * kDealK + k] = acc[k];
            }
        });
    }
    std::vector<int64_t> order(pieces.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return pieces[(size_t)a].count > pieces[(size_t)b].count; });
    std::vector<int64_t> load((size_t)n_tasks, 0), by_load((size_t)n_tasks);
    std::vector<float> tmom(balance_moments ? (size_t)n_tasks * kDealK : 0, 0.f);   // moments of every task so far
    std::vector<int64_t> task_piece((size_t)(n_tasks * RW), -1);    // [task][local row] -> piece
    for (int64_t lvl = 0; lvl * n_tasks < (int64_t)pieces.size(); ++lvl) {
        std::iota(by_load.begin(), by_load.end(), 0);
        std::stable_sort(by_load.begin(), by_load.end(), [&](int64_t a, int64_t b) { return load[(size_t)a] < load[(size_t)b]; });
        const int64_t lo = lvl * n_tasks, hi = std::min<int64_t>((int64_t)pieces.size(), lo + n_tasks);
        if (!balance_moments) {
            for (int64_t i = lo; i < hi; ++i) {
                const int64_t t = by_load[(size_t)(i - lo)];
                task_piece[(size_t)(t * RW + lvl)] = order[(size_t)i];
                load[(size_t)t] += pieces[(size_t)order[(size_t)i]].count;
            }
            continue;
        }
        const int64_t n_buckets = (hi - lo + kDealBucket - 1) / kDealBucket;
        parallel_for(n_buckets, [&](int64_t b_lo, int64_t b_hi) {
            std::vector<int64_t> ps, ts;
            for (int64_t b = b_lo; b < b_hi; ++b) {
                const int64_t i0 = lo + b * kDealBucket, i1 = std::min(hi, i0 + kDealBucket);
                ps.assign(order.begin() + i0, order.begin() + i1);                      // pieces of the bucket ...
                ts.assign(by_load.begin() + (i0 - lo), by_load.begin() + (i1 - lo));   // ... and its tasks
                // pieces with the largest moments choose first
                std::stable_sort(ps.begin(), ps.end(), [&](int64_t a, int64_t c) {
                    float na = 0.f, nc = 0.f;
                    for (int k = 0; k < kDealK; ++k) {
                        na += fabsf(mom[(size_t)a * kDealK + k]);
                        nc += fabsf(mom[(size_t)c * kDealK + k]);
                    }
                    return na > nc;
                });
                for (const int64_t pi : ps) {
                    const float *pm = &mom[(size_t)pi * kDealK];
                    size_t best = 0;
                    float best_cost = 3.4e38f;
                    for (size_t j = 0; j < ts.size(); ++j) {
                        const float *tm = &tmom[(size_t)ts[j] * kDealK];
                        float cost = 0.f;
                        for (int k = 0; k < kDealK; ++k) cost += (tm[k] + pm[k]) * (tm[k] + pm[k]);
                        if (cost < best_cost) {
                            best_cost = cost;
                            best = j;
                        }
                    }
                    const int64_t t = ts[best];
                    ts[best] = ts.back();
                    ts.pop_back();
                    task_piece[(size_t)(t * RW + lvl)] = pi;
                    load[(size_t)t] += pieces[(size_t)pi].count;
                    for (int k = 0; k < kDealK; ++k) tmom[(size_t)t * kDealK + k] += pm[k];
                }
            }
        }, 2);
    }
    // 2) column windows and the slot layout of every (task, window) bucket
    // 8 MiB windows; 16 MiB when the table slice is 128 MiB or more (C3 item rows, lead 2: 1.58 vs 1.61 ms; on the 25 MiB
    // table of the user rows 8 MiB: 1.53 vs 1.84 ms)
    int64_t win_kb = ngcf_opts().swept_window_kb > 0 ? ngcf_opts().swept_window_kb
                                                     : (((int64_t)col_hi - col_lo + 1) * kSW * 4 >= ((int64_t)128 << 20) ? 16384 : 8192);
    if (win_kb < 16) win_kb = 16;
    const int32_t win_cols = (int32_t)std::max<int64_t>(64, win_kb * 1024 / (kSW * 4));
    const int64_t n_win = ((int64_t)col_hi - col_lo) / win_cols + 1;
    if (n_tasks * n_win >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "swept plan: too many windows");
    std::vector<int64_t> tptr((size_t)(n_tasks * n_win) + 1, 0);
    auto bucket_hist = [&](int64_t t, std::vector<int32_t> &hist) {   // entries per (window, local row) of task t
        std::fill(hist.begin(), hist.end(), 0);
        for (int lr = 0; lr < RW; ++lr) {
            const int64_t pi = task_piece[(size_t)(t * RW + lr)];
            if (pi < 0) continue;
            const Piece &pc = pieces[(size_t)pi];
            for (int64_t x = pc.begin + pc.off; x < pc.end; x += pc.step)
                hist[(size_t)((col[(size_t)x] - col_lo) / win_cols) * RW + lr]++;
        }
    };
    // Layout of a (task, window) bucket, two ways (NGCF_SWEPT_ORDER):
    //  "rows"  : R = max(longest row, ceil(entries / 4)) rounds; row after row is laid down the rounds (wrap-around rule)
    //  "cols"  : entries sorted by column, rounds filled greedily in that order with entries of distinct rows (an entry
    //            whose row is already in the round waits for the next one): the wave walks the window left to right
    // Either way the entries of one row never share a round (they are read-modify-written in LDS without atomics).
    const bool by_cols = !ngcf_opts().swept_order_rows;   // default: cols
    struct Ent { int32_t col, lr; float v; };
    // schedule of one bucket: slot list (index into `ents`, -1 = empty), a multiple of kEPR long
    auto schedule_cols = [&](std::vector<Ent> &ents, std::vector<int32_t> &slots, std::vector<char> &done) {
        std::sort(ents.begin(), ents.end(), [](const Ent &x, const Ent &y) { return x.col != y.col ? x.col < y.col : x.lr < y.lr; });
        slots.clear();
        done.assign(ents.size(), 0);
        size_t first = 0, left = ents.size();
        while (left > 0) {
            uint64_t in_round[2] = {0, 0};
            int taken = 0;
            while (first < ents.size() && done[first]) ++first;
            for (size_t i = first; i < ents.size() && taken < kEPR && i < first + 512; ++i) {
                if (done[i]) continue;
                const int lr = ents[i].lr;
                if (in_round[lr >> 6] >> (lr & 63) & 1) continue;
                in_round[lr >> 6] |= (uint64_t)1 << (lr & 63);
                done[i] = 1;
                slots.push_back((int32_t)i);
                ++taken;
                --left;
            }
            for (; taken < kEPR; ++taken) slots.push_back(-1);
        }
    };
    auto gather_bucket = [&](int64_t t, std::vector<std::vector<Ent>> &per_win) {
        for (auto &v : per_win) v.clear();
        for (int lr = 0; lr < RW; ++lr) {
            const int64_t pi = task_piece[(size_t)(t * RW + lr)];
            if (pi < 0) continue;
            const Piece &pc = pieces[(size_t)pi];
            for (int64_t x = pc.begin + pc.off; x < pc.end; x += pc.step)
                per_win[(size_t)((col[(size_t)x] - col_lo) / win_cols)].push_back({col[(size_t)x], lr, val[(size_t)x]});
        }
    };
    parallel_for(n_tasks, [&](int64_t lo, int64_t hi) {
        std::vector<int32_t> hist((size_t)n_win * RW), slots;
        std::vector<std::vector<Ent>> per_win(by_cols ? (size_t)n_win : 0);
        std::vector<char> done;
        for (int64_t t = lo; t < hi; ++t) {
            if (by_cols) {
                gather_bucket(t, per_win);
                for (int64_t w = 0; w < n_win; ++w) {
                    schedule_cols(per_win[(size_t)w], slots, done);
                    tptr[(size_t)(t * n_win + w) + 1] = (int64_t)slots.size();
                }
                continue;
            }
            bucket_hist(t, hist);
            for (int64_t w = 0; w < n_win; ++w) {
                int64_t total = 0, longest = 0;
                for (int lr = 0; lr < RW; ++lr) {
                    total += hist[(size_t)w * RW + lr];
                    longest = std::max<int64_t>(longest, hist[(size_t)w * RW + lr]);
                }
                tptr[(size_t)(t * n_win + w) + 1] = std::max(longest, (total + kEPR - 1) / kEPR) * kEPR;
            }
        }
    });
    for (size_t i = 1; i < tptr.size(); ++i) tptr[i] += tptr[i - 1];
    const int64_t n_slots = tptr.back();
    // empty slots add 0 x (a row of their own window, an L2 hit) to the wave's spare accumulator row RW
    std::vector<int32_t> e_pack((size_t)std::max<int64_t>(n_slots, 1), -1);
    std::vector<float> e_val((size_t)std::max<int64_t>(n_slots, 1), 0.f);
    parallel_for(n_tasks, [&](int64_t lo, int64_t hi) {
        std::vector<int32_t> hist((size_t)n_win * RW), slots;
        std::vector<int64_t> rounds((size_t)n_win);
        std::vector<std::vector<Ent>> per_win(by_cols ? (size_t)n_win : 0);
        std::vector<char> done;
        for (int64_t t = lo; t < hi; ++t) {
            if (by_cols) {
                gather_bucket(t, per_win);
                for (int64_t w = 0; w < n_win; ++w) {
                    std::vector<Ent> &ents = per_win[(size_t)w];
                    schedule_cols(ents, slots, done);
                    const int64_t b0 = tptr[(size_t)(t * n_win + w)];
                    int32_t near_col = ents.empty() ? 0 : ents[0].col;
                    for (size_t i = 0; i < slots.size(); ++i) {
                        if (slots[i] >= 0) {
                            const Ent &e = ents[(size_t)slots[i]];
                            near_col = e.col;
                            e_pack[(size_t)b0 + i] = (int32_t)((uint32_t)e.lr << kRowBits) | e.col;
                            e_val[(size_t)b0 + i] = e.v;
                        } else {
                            e_pack[(size_t)b0 + i] = (int32_t)((uint32_t)RW << kRowBits) | near_col;   // spare row, a row just gathered
                        }
                    }
                }
                continue;
            }
            bucket_hist(t, hist);
            for (int64_t w = 0; w < n_win; ++w) {                   // hist -> first list position of every row in its bucket
                rounds[(size_t)w] = (tptr[(size_t)(t * n_win + w) + 1] - tptr[(size_t)(t * n_win + w)]) / kEPR;
                int32_t run = 0;
                for (int lr = 0; lr < RW; ++lr) {
                    const int32_t cnt = hist[(size_t)w * RW + lr];
                    hist[(size_t)w * RW + lr] = run;
                    run += cnt;
                }
            }
            for (int lr = 0; lr < RW; ++lr) {
                const int64_t pi = task_piece[(size_t)(t * RW + lr)];
                if (pi < 0) continue;
                const Piece &pc = pieces[(size_t)pi];
                for (int64_t x = pc.begin + pc.off; x < pc.end; x += pc.step) {
                    const int64_t w = (col[(size_t)x] - col_lo) / win_cols;
                    const int64_t k = hist[(size_t)w * RW + lr]++, R = rounds[(size_t)w];
                    const int64_t pos = tptr[(size_t)(t * n_win + w)] + (k % R) * kEPR + k / R;
                    e_pack[(size_t)pos] = (int32_t)((uint32_t)lr << kRowBits) | col[(size_t)x];
                    e_val[(size_t)pos] = val[(size_t)x];
                }
            }
            for (int64_t w = 0; w < n_win; ++w) {
                const int64_t b0 = tptr[(size_t)(t * n_win + w)], b1 = tptr[(size_t)(t * n_win + w) + 1];
                if (b0 == b1) continue;
                const int32_t spare = (int32_t)((uint32_t)RW << kRowBits) | (e_pack[(size_t)b0] & kColMask);   // slot 0 is never empty
                for (int64_t x = b0; x < b1; ++x)
                    if (e_pack[(size_t)x] < 0) e_pack[(size_t)x] = spare;
            }
        }
    });
    std::vector<int32_t> dst((size_t)(n_tasks * RW), -1), prow((size_t)(n_tasks * RW), 0);
    for (size_t i = 0; i < dst.size(); ++i) {
        if (task_piece[i] < 0) continue;
        const int64_t r = pieces[(size_t)task_piece[i]].row;
        dst[i] = r >= 0 ? (int32_t)(row_lo + r) : (int32_t)(-2 - (-1 - r));
        prow[i] = (int32_t)(row_lo + pieces[(size_t)task_piece[i]].src);   // the matrix row behind every accumulator row (edge dropout key)
    }
    if (ngcf_opts().swept_debug) {
        int64_t mx = 0, mn = INT64_MAX;
        for (int64_t t = 0; t < n_tasks; ++t) {
            const int64_t s = tptr[(size_t)((t + 1) * n_win)] - tptr[(size_t)(t * n_win)];
            mx = std::max(mx, s);
            mn = std::min(mn, s);
        }
        fprintf(stderr, "[swept plan] rows [%lld, %lld) nnz %lld cols [%d, %d]: %d waves x %d rows, %lld row passes, %lld windows x %d "
                        "cols, T %lld, pieces %zu (partial %lld), slots %lld (+%.1f%%), per task %lld..%lld\n",
                (long long)row_lo, (long long)row_hi, (long long)nnz, col_lo, col_hi, waves, RW, (long long)n_rowpass, (long long)n_win,
                win_cols, (long long)T, pieces.size(), (long long)n_partial, (long long)n_slots,
                100.0 * (double)(n_slots - nnz) / (double)nnz, (long long)mn, (long long)mx);
    }
    // 3) upload
    part.row_lo = row_lo;
    part.row_hi = row_hi;
    part.rows_per_wave = RW;
    part.lpe = LPE;
    part.n_rowpass = (int32_t)n_rowpass;
    part.n_win = (int32_t)n_win;
    part.win_cols = win_cols;
    part.col_lo = col_lo;
    part.col_hi = col_hi;
    part.n_slots = n_slots;
    part.n_partial = n_partial;
    part.n_heavy = (int64_t)heavy_row.size();
    int rc = upload(&part.tptr, tptr, stream);
    if (rc == NGCF_OK) rc = upload(&part.e_pack, e_pack, stream);
    if (rc == NGCF_OK) rc = upload(&part.e_val, e_val, stream);
    if (rc == NGCF_OK) rc = upload(&part.dst, dst, stream);
    if (rc == NGCF_OK) rc = upload(&part.prow, prow, stream);
    if (rc == NGCF_OK && part.n_heavy > 0) rc = upload(&part.heavy_row, heavy_row, stream);
    if (rc == NGCF_OK && part.n_heavy > 0) rc = upload(&part.heavy_seg_ptr, heavy_ptr, stream);
    if (rc == NGCF_OK && hipStreamSynchronize(stream) != hipSuccess) rc = fail(NGCF_ERR_HIP, "swept plan: upload failed");
    if (rc != NGCF_OK) {
        free_part(part);
        return rc;
    }
    part.waves = waves;
    return NGCF_OK;
}
}  // namespace

void free_swept(ngcf_csr *c)
{
    ngcf_csr::Swept &w = c->swept;
    for (auto &p : w.parts) free_part(p);
    free_segset(w.out);
    if (w.barrier) (void)hipFree(w.barrier);
    w = ngcf_csr::Swept();
}

// Builds the parts for c->mode (2: every row group whose shape allows it, 3: the groups where re-use is expected) and
// the segment set of the cut rows that stay with the row-wise kernels.  Workspace rows: [those segments][part 0's
// partial rows][part 1's]...
int build_swept_plan(ngcf_csr *c, hipStream_t stream)
{
    free_swept(c);
    ngcf_csr::Swept &w = c->swept;
    w.built_mode = c->mode;
    w.lpe = ngcf_opts().swept_lpe == 32 ? 32 : 16;
    if (c->mode < 2 || c->n_rows == 0 || c->nnz == 0) return NGCF_OK;
    // the plan is laid out for one resident workgroup on each of 256 CUs in 8 XCDs (an MI355X in SPX mode); on any
    // other partitioning the grid would not be resident together, so the products stay on the row-wise kernels
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus != kSweptWGs) {
        static bool warned = false;
        if (!warned) {
            warned = true;
            fprintf(stderr, "[ngcf] note: this device reports %d CUs, not %d (an MI355X in SPX mode): the L2-swept SpMM plan is not built and the "
                            "products run on the row-wise kernels (about half the speed on large graphs); ngcf_csr_swept_rows() returns 0\n",
                    cus, kSweptWGs);
        }
        return NGCF_OK;
    }
    w.group_swept.assign(c->groups.size(), 0);
    for (size_t g = 0; g < c->groups.size(); ++g) {
        ngcf_csr::Swept::Part p;
        const int rc = build_part(c, c->groups[g], c->mode == 2, stream, p, w.lpe);
        if (rc != NGCF_OK) {
            free_swept(c);
            return rc;
        }
        if (p.waves == 0) continue;
        w.group_swept[g] = 1;
        w.parts.push_back(p);
    }
    if (w.parts.empty()) return NGCF_OK;
    std::vector<int64_t> rp((size_t)c->n_rows + 1);
    HIP_TRY(hipMemcpyAsync(rp.data(), c->rowptr, sizeof(int64_t) * rp.size(), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::vector<int32_t> seg_row, heavy_row;
    std::vector<int64_t> seg_begin, heavy_ptr(1, 0);
    for (size_t g = 0; g < c->groups.size(); ++g) {
        if (w.group_swept[g]) continue;
        for (int64_t r = c->groups[g].begin; r < c->groups[g].end; ++r) {
            if (rp[(size_t)r + 1] - rp[(size_t)r] <= c->seg_len) continue;
            heavy_row.push_back((int32_t)r);
            for (int64_t b = rp[(size_t)r]; b < rp[(size_t)r + 1]; b += c->seg_len) {
                seg_row.push_back((int32_t)r);
                seg_begin.push_back(b);
            }
            heavy_ptr.push_back((int64_t)seg_row.size());
        }
    }
    w.out.n_seg = (int64_t)seg_row.size();
    w.out.n_heavy = (int64_t)heavy_row.size();
    int64_t base = w.out.n_seg;
    for (auto &p : w.parts) {
        p.partial_base = base;
        base += p.n_partial;
        w.n_partial += p.n_partial;
    }
    int rc = NGCF_OK;
    if (w.out.n_seg > 0) {
        rc = upload(&w.out.seg_row, seg_row, stream);
        if (rc == NGCF_OK) rc = upload(&w.out.seg_begin, seg_begin, stream);
        if (rc == NGCF_OK) rc = upload(&w.out.heavy_row, heavy_row, stream);
        if (rc == NGCF_OK) rc = upload(&w.out.heavy_seg_ptr, heavy_ptr, stream);
    }
    if (rc == NGCF_OK && (hipMalloc(&w.barrier, sizeof(uint32_t) * 32 * 8 * 4) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess))
        rc = fail(NGCF_ERR_HIP, "swept plan: allocation failed");
    if (rc != NGCF_OK) free_swept(c);
    return rc;
}

// The parts gather with 32-bit byte offsets: the gathered rows must lie within 4 GiB of the table base
bool swept_usable(const ngcf_csr *c, int64_t ldE, int d)
{
    const ngcf_csr::Swept &w = c->swept;
    if (w.parts.empty() || c->mode < 2 || d % (w.lpe * 4) != 0) return false;
    for (const auto &p : w.parts)
        if (((int64_t)p.col_hi + 1) * ldE * 4 > (int64_t)UINT32_MAX) return false;
    return true;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef NGCF_SWEPT_LABW
#define NGCF_SWEPT_LABW 2      // loads allowed in flight when the adds of a half-chunk begin (see load_entries below)
#endif

// RW accumulator rows per wave, NW waves per workgroup (NW*RW*256 B of LDS)
// DBG: every kDbgEvery chunks a wave stores (s_memrealtime, column it is gathering) into `dbg` - the sweep-spread
// trace of tools/swept_trace_lab.py (NGCF_SWEPT_TRACE=<file>); the product never runs this instantiation otherwise
constexpr int kDbgEvery = 2, kDbgSamples = 512;
// One launch can take up to kMaxLaunchParts parts (row groups) of a product one after the other: a workgroup that has finished
// its share of the first goes straight on to the second, so the end-of-part stragglers (the last waves finish 5-8 % after the
// first, profiles/r02_swept_trace.txt) would cost once per product instead of once per part; the sweep steps are numbered
// through, so the XCD counters need no reset in between.  Off by default (launch_swept: measured slower).
constexpr int kMaxLaunchParts = 4;
struct SweptPartArgs {
    const int64_t *tptr;
    const int32_t *e_pack;
    const float *e_val;
    const int32_t *dst;
    const int32_t *prow;
    float *partial;
    int n_rowpass, n_win, lead;
};
struct SweptLaunch {
    SweptPartArgs part[kMaxLaunchParts];
    int n_parts;
};

// DROP: device-mode node dropout (common.h, EdgeDrop): a lane tests the entry it holds when it loads it - row from the wave's
// table of matrix rows (one ds_bpermute), column from the packed entry - and zeroes the value of a dropped entry; the gather of
// a dropped entry still happens (the round structure is fixed by the plan), its product is 0.
template <int LPE, int RW, int NW, bool DBG, bool DROP>
__global__ __launch_bounds__(NW * 64) void spmm_swept_kernel(SweptLaunch L, int n_slices, const float *__restrict__ E,
                                                             int64_t ldE, float *__restrict__ out, int64_t ldo, int dp, unsigned *bar,
                                                             int max_spin, int sync_k, unsigned prio_cols, int prio_graded, int nt_flags,
                                                             unsigned long long *__restrict__ dbg, EdgeDrop dr_in)
{
    const EdgeDropR dr = resolve_drop(dr_in);
    constexpr int kSW = LPE * 4, kEPR = 64 / LPE, kCH = 16 * kEPR;
    __shared__ float acc_lds[NW * (RW + 1) * kSW];   // per wave: RW accumulator rows + the spare row of the empty slots
    __shared__ unsigned wg_cnt[kRing];
    __shared__ int perm_lds;                    // highest sweep step this workgroup knows to be permitted
    __shared__ unsigned xcc_id;
    __shared__ unsigned wg_front;               // furthest (sweep, column) any wave of this workgroup has reached
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane / LPE, p = lane % LPE;   // entry slot in the round, position in the slice
    // chunk entry this lane keeps: round u = lane u of every 16-lane DPP row (LPE = 32: the two DPP rows of an entry's lanes hold it twice)
    const int held = (lane & 15) * kEPR + g;
    float *wacc = acc_lds + wave * ((RW + 1) * kSW);
    if (threadIdx.x < kRing) wg_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;   // HW_REG_XCC_ID
        perm_lds = L.part[0].lead;              // steps 0..lead wait for nobody
        wg_front = 0;
    }
    __syncthreads();
    unsigned *ctr = bar + xcc_id * 32;
    const unsigned members = gridDim.x / 8;
    int perm = L.part[0].lead;                  // this wave's copy of perm_lds
    int step0 = 0;                              // sweep steps of the passes before this one
    const unsigned ld_bytes = (unsigned)ldE * 4u;
    int dbg_n = 0, dbg_chunk = 0;
    unsigned long long *dbg_w = DBG ? dbg + ((size_t)blockIdx.x * NW + wave) * (2 * kDbgSamples + 2) : nullptr;
    if (DBG && lane == 0) dbg_w[0] = xcc_id;
    int sweep_no = 0;                           // sweeps (row pass x slice) so far, over all parts
    for (int pi = 0; pi < L.n_parts; ++pi) {
    const int64_t *__restrict__ tptr = L.part[pi].tptr;
    const int32_t *__restrict__ e_pack = L.part[pi].e_pack;
    const float *__restrict__ e_val = L.part[pi].e_val;
    const int32_t *__restrict__ dst = L.part[pi].dst;
    float *__restrict__ partial = L.part[pi].partial;
    const int n_rowpass = L.part[pi].n_rowpass, n_win = L.part[pi].n_win, lead = L.part[pi].lead;
    const bool sync = lead >= 0;
    for (int rp = 0; rp < n_rowpass; ++rp) {
        const int64_t task = ((int64_t)rp * gridDim.x + blockIdx.x) * NW + wave;
        const int64_t *tp = tptr + task * n_win;
        const int64_t beg = tp[0], end = tp[n_win];
        int my_row = 0, my_row_hi = 0;           // lane lr (and lr - 64): the matrix row behind accumulator row lr of this task
        if (DROP && lane < RW) my_row = L.part[pi].prow[task * RW + lane];
        if (DROP && RW > 64 && lane + 64 < RW) my_row_hi = L.part[pi].prow[task * RW + 64 + lane];
        for (int slice = 0; slice < n_slices; ++slice, ++sweep_no, step0 += (n_win + sync_k - 1) / sync_k) {   // sweep step = sync_k windows
            const char *Eb = reinterpret_cast<const char *>(E + slice * kSW);   // uniform base + 32-bit lane offsets
            for (int i = lane; i < (RW + 1) * kSW; i += 64) wacc[i] = 0.f;
            int b = 0;
            int64_t wend = tp[1], wend_next = n_win > 1 ? tp[2] : end;
            auto arrive = [&](int s) {           // this wave has left sweep step s behind
                if (sync && lane == 0) {
                    const unsigned old = atomicAdd(&wg_cnt[s % kRing], 1u);
                    if ((old + 1) % NW == 0)
                        __hip_atomic_fetch_add(ctr + (s % kRing), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            // windows left behind by a wave whose next slot is `pos`; then the permission to enter the new one
            auto cross = [&](int64_t pos, bool crossed) {
                while (b < n_win - 1 && pos >= wend) {
                    if ((b + 1) % sync_k == 0) arrive(step0 + b / sync_k);
                    ++b;
                    wend = wend_next;
                    wend_next = b + 2 <= n_win ? tp[b + 2] : end;
                    crossed = true;
                }
                const int s = step0 + b / sync_k;   // the step being entered needs step s-1-lead finished by the whole XCD
                if (crossed && sync && max_spin > 0 && perm < s) {
                    perm = __hip_atomic_load(&perm_lds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    int spins = 0;
                    while (perm < s) {
                        // lanes 0..kRing-1 read the ring; step t is finished when its slot reached members*(t/kRing+1)
                        const unsigned mine = lane < kRing ? __hip_atomic_load(ctr + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                        int t = perm - lead;     // first step not yet known to be finished
                        for (int k = 0; k < kRing - 2 - lead; ++k, ++t) {
                            const unsigned have = __builtin_amdgcn_readlane(mine, t % kRing);
                            if (have < members * (unsigned)(t / kRing + 1)) break;
                        }
                        perm = t + lead;         // steps <= perm may start
                        if (perm >= s) break;
                        if (++spins >= max_spin) {
                            max_spin = 0;        // the XCD's workgroups are not resident together: stop waiting for good
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (lane == 0) atomicMax(&perm_lds, perm);
                }
            };
            const int idle_pk = end > beg ? (RW << kRowBits) | (e_pack[beg] & kColMask) : 0;   // past the end: like an empty slot
            // Every load of the main loop is UNCONDITIONAL (slots past the end of the list read the list's last entry again and
            // are turned into idle slots afterwards): behind a branch the compiler cannot count a load and waits for everything
            // in flight instead - which is what r01/r02 ran with unknowingly (s_waitcnt vmcnt(7..0) in front of the adds of a
            // half: at most the other half's eight gathers in flight).  With exact counts (vmcnt(15..8): both halves in flight)
            // the product is SLOWER - 3.16 vs 2.96-2.99 ms per L.E on C3 - because waves that run further ahead of their
            // data spread the sweep; so the depth is now stated: all but NGCF_SWEPT_LABW loads have arrived before the adds
            // of a half begin.  C3, ms per L.E, same box: 0 -> 2.99-3.00, 2 -> 2.94, 3 -> 2.95, 4 -> 2.95 (3.05 once), 5 -> 2.96,
            // 6 -> 2.97, 8 -> 3.16, 12 -> 3.13, no explicit wait -> 3.16-3.18 (0..6 are within run-to-run noise of each other: +-0.04).
            auto load_entries = [&](int64_t pos, int &pk, float &v) {
                const int64_t idx = pos + held;
                const int64_t idc = idx < end ? idx : end - 1;
                int pk_l;
                float v_l;
#ifdef NGCF_LAB
                if (nt_flags & 1) {              // the entry lists are read once: streaming loads, they should not displace table rows in L2
                    pk_l = __builtin_nontemporal_load(&e_pack[idc]);
                    v_l = __builtin_nontemporal_load(&e_val[idc]);
                } else
#endif
                {
                    pk_l = e_pack[idc];
                    v_l = e_val[idc];
                }
                pk = idx < end ? pk_l : idle_pk;
                v = idx < end ? v_l : 0.f;
                if (DROP) {
                    const int lr = (pk >> kRowBits) & 0x7f;                      // RW (the spare row) for idle slots: any row will do
                    int grow = __shfl(my_row, lr < RW ? lr & 63 : 0);
                    if (RW > 64) {
                        const int hi = __shfl(my_row_hi, lr & 63);
                        grow = lr >= 64 && lr < RW ? hi : grow;
                    }
                    v = edge_keep(dr, grow, pk & kColMask) ? v : 0.f;
                }
            };
            if (end > beg) {
                int pkA, pkB;
                float vA, vB;
                f32x4 xa[8], xb[8];
                unsigned offA;
                const unsigned lane_off = p * 16;
#define NGCF_GATHER(buf, u, j) buf[j] = *reinterpret_cast<const f32x4 *>(Eb + ((unsigned)row_bcast<u>((int)offA) + lane_off));
#define NGCF_GATHER_LO(x) NGCF_GATHER(x, 0, 0) NGCF_GATHER(x, 1, 1) NGCF_GATHER(x, 2, 2) NGCF_GATHER(x, 3, 3) NGCF_GATHER(x, 4, 4) NGCF_GATHER(x, 5, 5) NGCF_GATHER(x, 6, 6) NGCF_GATHER(x, 7, 7)
#define NGCF_GATHER_HI(x) NGCF_GATHER(x, 8, 0) NGCF_GATHER(x, 9, 1) NGCF_GATHER(x, 10, 2) NGCF_GATHER(x, 11, 3) NGCF_GATHER(x, 12, 4) NGCF_GATHER(x, 13, 5) NGCF_GATHER(x, 14, 6) NGCF_GATHER(x, 15, 7)
#define NGCF_ACC(buf, u, j)                                                                        \
    {                                                                                             \
        const int pku = row_bcast<u>(pkA);                                                        \
        const float vu = row_bcast<u>(vA);                                                        \
        if (pku >= 0) {                                                                           \
            f32x4 *a = reinterpret_cast<f32x4 *>(wacc + (pku >> kRowBits) * kSW + p * 4);         \
            f32x4 t = *a;                                                                         \
            t.x = fmaf(vu, buf[j].x, t.x);                                                          \
            t.y = fmaf(vu, buf[j].y, t.y);                                                          \
            t.z = fmaf(vu, buf[j].z, t.z);                                                          \
            t.w = fmaf(vu, buf[j].w, t.w);                                                          \
            *a = t;                                                                               \
        }                                                                                         \
    }
#define NGCF_ACC_LO(x) NGCF_ACC(x, 0, 0) NGCF_ACC(x, 1, 1) NGCF_ACC(x, 2, 2) NGCF_ACC(x, 3, 3) NGCF_ACC(x, 4, 4) NGCF_ACC(x, 5, 5) NGCF_ACC(x, 6, 6) NGCF_ACC(x, 7, 7)
#define NGCF_ACC_HI(x) NGCF_ACC(x, 8, 0) NGCF_ACC(x, 9, 1) NGCF_ACC(x, 10, 2) NGCF_ACC(x, 11, 3) NGCF_ACC(x, 12, 4) NGCF_ACC(x, 13, 5) NGCF_ACC(x, 14, 6) NGCF_ACC(x, 15, 7)
                load_entries(beg, pkA, vA);
                cross(beg, true);
                offA = (unsigned)(pkA & kColMask) * ld_bytes;
                NGCF_GATHER_LO(xa)
                const unsigned sweep_tag = (unsigned)(sweep_no & 0xff) << kRowBits;
                for (int64_t pos = beg; pos < end; pos += kCH) {
                    if (prio_cols) {
                        // Inside a workgroup the same few waves fall behind and stay behind (the instruction arbiter serves the
                        // older wave of a SIMD first), up to 4.5 MiB of table on the C3 item rows while the workgroups themselves
                        // stay within 0.3 MiB of each other (tools/swept_trace_lab.py): a wave that is more than prio_cols
                        // columns behind the front of its workgroup raises its priority, the others run at 0.
                        const unsigned mine = sweep_tag | (unsigned)(__builtin_amdgcn_readfirstlane(pkA) & kColMask);
                        unsigned front = 0;
                        if (lane == 0) front = atomicMax(&wg_front, mine);
                        front = (unsigned)__builtin_amdgcn_readfirstlane((int)front);
                        const unsigned lag = front > mine ? front - mine : 0u;
                        if (prio_graded) {
                            if (lag > 4 * prio_cols) __builtin_amdgcn_s_setprio(3);
                            else if (lag > 2 * prio_cols) __builtin_amdgcn_s_setprio(2);
                            else if (lag > prio_cols) __builtin_amdgcn_s_setprio(1);
                            else __builtin_amdgcn_s_setprio(0);
                        } else {
                            if (lag > prio_cols) __builtin_amdgcn_s_setprio(3);
                            else __builtin_amdgcn_s_setprio(0);
                        }
                    }
                    if (DBG) {
                        if (dbg_chunk++ % kDbgEvery == 0 && dbg_n < kDbgSamples && lane == 0) {
                            dbg_w[2 + 2 * dbg_n] = __builtin_amdgcn_s_memrealtime();
                            dbg_w[3 + 2 * dbg_n] = ((unsigned long long)sweep_no << 32) | (unsigned)(pkA & kColMask);
                            ++dbg_n;
                        }
                    }
                    load_entries(pos + kCH, pkB, vB);                // the next chunk's entries, two gathers ahead
                    if (pos + 8 * kEPR < end) cross(pos + 8 * kEPR, false);   // no gather is issued into a window before its permission
                    NGCF_GATHER_HI(xb)                               // (past the end: idle slots, the column of the task's first entry)
                    __builtin_amdgcn_s_waitcnt(0xF70 | (NGCF_SWEPT_LABW & 0xf));   // s_waitcnt vmcnt(NGCF_SWEPT_LABW)
                    NGCF_ACC_LO(xa)
                    if (pos + kCH < end) cross(pos + kCH, false);
                    offA = (unsigned)(pkB & kColMask) * ld_bytes;
                    NGCF_GATHER_LO(xa)                               // first half of the next chunk, in flight during the adds below
                    __builtin_amdgcn_s_waitcnt(0xF70 | (NGCF_SWEPT_LABW & 0xf));
                    NGCF_ACC_HI(xb)
                    pkA = pkB;
                    vA = vB;
                }
#undef NGCF_GATHER
#undef NGCF_GATHER_LO
#undef NGCF_GATHER_HI
#undef NGCF_ACC
#undef NGCF_ACC_LO
#undef NGCF_ACC_HI
            }
            cross(INT64_MAX - 1, false);         // leave the remaining windows (b ends at n_win-1) ...
            arrive(step0 + (n_win - 1) / sync_k);   // ... and the last step
            // write this wave's rows (its own LDS rows; a wave's LDS operations complete in order)
            for (int r0 = 0; r0 < RW; r0 += kEPR) {
                const int r = r0 + g;
                if (r < RW) {
                    const int drow = dst[task * RW + r];
                    if (drow != -1) {
                        float *o = drow >= 0 ? out + (int64_t)drow * ldo : partial + (int64_t)(-2 - drow) * dp;
                        const f32x4 res = *reinterpret_cast<const f32x4 *>(wacc + r * kSW + p * 4);
#ifdef NGCF_LAB
                        if (nt_flags & 2) __builtin_nontemporal_store(res, reinterpret_cast<f32x4 *>(o + slice * kSW + p * 4));
                        else
#endif
                        *reinterpret_cast<f32x4 *>(o + slice * kSW + p * 4) = res;
                    }
                }
            }
        }
    }
    }   // parts
    if (DBG && lane == 0) dbg_w[1] = (unsigned long long)dbg_n;
}

// kernels + fix-ups of every part; `partial` = workspace base (rows of dp floats)
int launch_swept(const ngcf_csr *c, const float *E, int64_t ldE, int d, float *out, int64_t ldo, float *partial, int dp,
                 hipStream_t stream, const EdgeDrop &dr)
{
    const ngcf_csr::Swept &w = c->swept;
    const NgcfOptions &o = ngcf_opts();
    const int max_spin = o.swept_spin;                       // polls before a wave stops waiting for good
    // Inside a window a wave's entries are in ascending column order (plan, "cols" layout), so waves that enter a window
    // together walk the table left to right together and the live set is a few hundred KiB; the XCD-wide counters tick
    // once per 8 MiB window and a wave may be one window ahead - they only bound the drift.  bench.py on C3, same process:
    // row-wise layout inside 4 MiB windows 15.1 ms/step (SpMM 4.16 ms), inside 2 MiB windows with a tick every third
    // 14.3-14.4 (3.9); column order inside 8 MiB windows 13.65-13.7 (3.68-3.71), 6 MiB 13.8, 12 MiB 13.9, 16 MiB 14.3,
    // 4 MiB 14.4; lead 0: 14.9, lead 2: 14.5.
    // r02, with the graded wave priorities below (profiles/r02_swept_lab.txt, grid of window x lead x threshold): a table of
    // many windows wants lead 2 (C3 item rows, 16 windows: 1.50 vs 1.53 ms), a table of a few windows lead 1 (C3 user rows, 4
    // windows: 1.44 vs 1.46 ms); NGCF_SWEPT_LEAD overrides both (-1: no synchronisation at all)
    const int lead_env = o.swept_lead;
    const int sync_k = std::max(1, o.swept_sync_every);       // windows per sweep step (lab knob)
    // lag (KiB of table slice) behind the front of its workgroup beyond which a wave raises its priority; 0 = off
    const int kSW = w.lpe * 4;
    const unsigned prio_cols = (unsigned)std::max(0, o.swept_prio_kb) * 1024u / (kSW * 4);
    // graded: priority 1 / 2 / 3 beyond 1x / 2x / 4x the threshold (C3 item rows 1.50 vs 1.61 ms against one step to 3)
    const int prio_graded = o.swept_prio_graded;
#ifdef NGCF_LAB
    const int nt_flags = o.swept_nt;                                           // 1: streaming loads of the entry lists, 2: streaming stores of the rows
    const char *trace = o.swept_trace[0] ? o.swept_trace : nullptr;
#else
    const int nt_flags = 0;
#endif
    // consecutive parts of the same workgroup shape share a launch
    // one launch per part by default: taking both halves of C3 in one launch (NGCF_SWEPT_MERGE=1) was measured SLOWER (3.01-3.11
    // vs 2.96-2.98 ms per product, same box) - the workgroups that start the second part early gather from another table
    // and take L2 away from the stragglers of the first
    #ifdef NGCF_LAB
    const int max_parts = o.swept_merge ? kMaxLaunchParts : 1;
#else
    const int max_parts = 1;
#endif
    for (size_t p0 = 0; p0 < w.parts.size();) {
        SweptLaunch L{};
        size_t p1 = p0;
        for (; p1 < w.parts.size() && L.n_parts < max_parts && w.parts[p1].waves == w.parts[p0].waves; ++p1) {
            const auto &p = w.parts[p1];
            // sweep steps a wave may run ahead: by the size of the part's table
            const int lead = std::min(lead_env != -2 ? lead_env : (p.n_win >= 8 ? 2 : 1), kRing - 4);
            L.part[L.n_parts++] = SweptPartArgs{p.tptr, p.e_pack, p.e_val, p.dst, p.prow, partial ? partial + p.partial_base * (int64_t)dp : nullptr,
                                                p.n_rowpass, p.n_win, lead};
        }
        const int waves = w.parts[p0].waves;
        // this stream's block of sweep counters (up to four streams per CSR get one of their own; a fifth shares the last)
        int bi = 0;
        for (; bi < w.barrier_used && w.barrier_owner[bi] != stream; ++bi) {}
        if (bi == w.barrier_used) {
            if (w.barrier_used < 4) w.barrier_owner[w.barrier_used++] = stream;
            else bi = 3;
        }
        unsigned *bar_blk = w.barrier + (size_t)bi * 32 * 8;
        HIP_TRY(hipMemsetAsync(bar_blk, 0, sizeof(uint32_t) * 32 * 8, stream));
        unsigned long long *dbg = nullptr;
#ifdef NGCF_LAB
        const size_t dbg_words = (size_t)kSweptWGs * waves * (2 * kDbgSamples + 2);
        if (trace) {
            HIP_TRY(hipMalloc(&dbg, dbg_words * 8));
            HIP_TRY(hipMemsetAsync(dbg, 0, dbg_words * 8, stream));
        }
#endif
#define NGCF_SWEPT_LAUNCH(LPE_, RW_, NW_, DBG_, DROP_)                                                                                    \
    spmm_swept_kernel<LPE_, RW_, NW_, DBG_, DROP_><<<dim3(kSweptWGs), NW_ * 64, 0, stream>>>(L, d / kSW, E, ldE, out, ldo, dp, bar_blk, \
                                                                                      max_spin, sync_k, prio_cols, prio_graded,    \
                                                                                      nt_flags, dbg, dr)
#ifdef NGCF_LAB
        if (waves == 8) {
            if (dr.n > 0) NGCF_SWEPT_LAUNCH(16, 72, 8, false, true);
            else if (trace) NGCF_SWEPT_LAUNCH(16, 72, 8, true, false);
            else NGCF_SWEPT_LAUNCH(16, 72, 8, false, false);
        } else if (trace && dr.n == 0) {
            NGCF_SWEPT_LAUNCH(16, 36, 16, true, false);
        } else
#endif
        if (waves != 16) return fail(NGCF_ERR_ARG, "swept: the plan's workgroup shape (%d waves) is not compiled into this library", waves);
        else if (w.lpe == 32 && dr.n > 0) NGCF_SWEPT_LAUNCH(32, 18, 16, false, true);
        else if (w.lpe == 32) NGCF_SWEPT_LAUNCH(32, 18, 16, false, false);
        else if (dr.n > 0) NGCF_SWEPT_LAUNCH(16, 36, 16, false, true);
        else NGCF_SWEPT_LAUNCH(16, 36, 16, false, false);
#undef NGCF_SWEPT_LAUNCH
        LAUNCH_CHECK();
#ifdef NGCF_LAB
        if (trace) {     // lab only: host-synchronous dump, one file per launch
            const auto &p = w.parts[p0];
            std::vector<unsigned long long> h(dbg_words);
            HIP_TRY(hipStreamSynchronize(stream));
            HIP_TRY(hipMemcpy(h.data(), dbg, dbg_words * 8, hipMemcpyDeviceToHost));
            (void)hipFree(dbg);
            char path[512];
            snprintf(path, sizeof(path), "%s.part%d", trace, (int)p0);
            if (FILE *f = fopen(path, "wb")) {
                const long long hdr[6] = {kSweptWGs, p.waves, kDbgSamples, p.win_cols, p.n_win, p.col_lo};
                fwrite(hdr, sizeof(hdr), 1, f);
                fwrite(h.data(), 8, h.size(), f);
                fclose(f);
            }
        }
#endif
        p0 = p1;
    }
    for (const auto &p : w.parts) {
        if (p.n_heavy == 0) continue;
        spmm_fixup_kernel<4><<<dim3((unsigned)((p.n_heavy + 3) / 4)), 256, 0, stream>>>(
            p.heavy_row, p.heavy_seg_ptr, p.n_heavy, partial + p.partial_base * (int64_t)dp, dp, d, out, ldo);
        LAUNCH_CHECK();
    }
    return NGCF_OK;
}
