// hostrng.hip - HOST code: the dropout masks of the reference drawn from torch's default CPU generator, at SIMD speed (r04).
//
// The reference draws its node- and message-dropout masks with `nn.Dropout` on CPU tensors (NGCF.py:93-100,142): torch's CPU
// `bernoulli_(keep)` walks the tensor serially and takes, per element, one 64-bit draw of the default generator (mt19937: two
// 32-bit outputs, the first one the high half), keeps its low 53 bits as x = v * 2^-53 and sets the element to (x < keep)
// (ATen: bernoulli_scalar_kernel_default -> bernoulli_distribution<double> -> uniform_real_distribution<double> -> random64()).
// "reference" mode (NGCF.node_dropout_mode / mess_dropout_mode, the defaults) reproduces those masks bit for bit, so the stream
// has to be THAT stream - mt19937 has no cheap jump-ahead, so it is generated on the host - but nothing says it has to be
// generated one element at a time behind a mutex: here the 624-word state is regenerated and tempered in vector loops and the
// comparison runs on integers.  The caller hands over the bytes of `torch.get_rng_state()`, gets the flags / the noise tensor and
// the advanced state back and installs it with `torch.set_rng_state` - the default generator ends up exactly where the reference
// would have left it.  The Python side checks this routine against torch itself once per process before it trusts it.
#include "common.h"

#include <cstdint>
#include <cstring>

namespace {
constexpr int kN = 624, kM = 397;

// layout of at::CPUGeneratorImplState (legacy POD first): uint64 seed; int left; int seeded; uint64 next; uint64 state[624]; ...
constexpr size_t kOffLeft = 8, kOffSeeded = 12, kOffNext = 16, kOffState = 24, kMinBytes = kOffState + 8 * kN;

__attribute__((always_inline)) inline uint32_t twist(uint32_t u, uint32_t v)
{
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}

// one regeneration of the state (at::mt19937::next_state), written as three loops without loop-carried dependencies
__attribute__((always_inline)) inline void regenerate(uint32_t *__restrict__ s, uint32_t *__restrict__ t)
{
    for (int j = 0; j < kN - kM; ++j) t[j] = s[j + kM] ^ twist(s[j], s[j + 1]);                    // 0 .. 226: old words only
    for (int j = kN - kM; j < 2 * (kN - kM); ++j) t[j] = t[j - (kN - kM)] ^ twist(s[j], s[j + 1]);  // 227 .. 453: new words 0 .. 226
    for (int j = 2 * (kN - kM); j < kN - 1; ++j) t[j] = t[j - (kN - kM)] ^ twist(s[j], s[j + 1]);   // 454 .. 622: new words 227 .. 395
    t[kN - 1] = t[kM - 1] ^ twist(s[kN - 1], t[0]);
    memcpy(s, t, sizeof(uint32_t) * kN);
}

__attribute__((always_inline)) inline uint32_t temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
// the elements of one block: pairs of tempered words -> kept or not.  All integer: x = v * 2^-53 < keep  <=>  v < keep * 2^53
// <=> v < T with T = ceil(keep * 2^53) (v is an integer; the product is exact, a power-of-two scaling).  No u64 -> double conversion
// (AVX2 has none, AVX-512F neither), no branch in the loop body: FLAGS / NOISE are compile-time.
template <bool FLAGS, bool NOISE>
__attribute__((always_inline)) inline int64_t classify(const uint32_t *__restrict__ w, int pairs, uint64_t T, uint8_t *__restrict__ flags,
                                                       float *__restrict__ noise, float scale)
{
    int64_t kept = 0;
    for (int k = 0; k < pairs; ++k) {                          // first word = high half (make64BitsFrom32Bits(random1, random2))
        const uint64_t v = ((uint64_t)(w[2 * k] & 0x1fffffu) << 32) | w[2 * k + 1];
        const int keep_it = v < T;
        kept += keep_it;
        if (FLAGS) flags[k] = (uint8_t)keep_it;
        if (NOISE) noise[k] = keep_it ? scale : 0.f;
    }
    return kept;
}
}  // namespace

// n Bernoulli(keep) draws from the generator whose state bytes are `rng_state` (torch.get_rng_state(), updated in place):
// flags[i] = 1 if kept (may be NULL), noise[i] = kept ? scale : 0 (may be NULL), *n_kept = number kept.
// (compiled three times - AVX-512, AVX2, baseline - and picked at load time: the loops above and below are plain vector loops)
extern "C" __attribute__((target_clones("avx512f", "avx2", "default"))) int ngcf_torch_cpu_bernoulli(uint8_t *rng_state, int64_t state_bytes, int64_t n, double keep, uint8_t *flags, float *noise,
                                        float scale, int64_t *n_kept)
{
    if (!rng_state || n < 0 || (size_t)state_bytes < kMinBytes) return fail(NGCF_ERR_ARG, "torch_cpu_bernoulli: bad argument");
    int32_t left, seeded;
    uint64_t next;
    memcpy(&left, rng_state + kOffLeft, 4);
    memcpy(&seeded, rng_state + kOffSeeded, 4);
    memcpy(&next, rng_state + kOffNext, 8);
    // at::mt19937: every draw does `if (--left == 0) next_state()` then reads state[next++]; next_state sets left = 624, next = 0
    if (!(left >= 1 && left <= kN && next <= (uint64_t)kN && (left + (int64_t)next == kN + 1 || (left == 1 && next == 0))))
        return fail(NGCF_ERR_ARG, "torch_cpu_bernoulli: unexpected generator state (left %d, next %llu)", left, (unsigned long long)next);
    uint32_t s[kN], t[kN];
    for (int j = 0; j < kN; ++j) {
        uint64_t w;
        memcpy(&w, rng_state + kOffState + 8 * (size_t)j, 8);
        s[j] = (uint32_t)w;
    }
    if (!(keep >= 0.0 && keep <= 1.0)) return fail(NGCF_ERR_ARG, "torch_cpu_bernoulli: keep probability %g not in [0, 1]", keep);
    const double thresh = keep * 9007199254740992.0;           // keep * 2^53: exact (a power of two), so x < keep <=> v < thresh
    uint64_t T = (uint64_t)thresh;
    if ((double)T < thresh) ++T;                               // ceil: v < thresh <=> v < T for integer v
    int pos = left == 1 ? kN : (int)next;                      // index of the next word; kN = "regenerate first"
    int64_t kept = 0, i = 0;
    uint32_t w[kN + 2];                                        // tempered words of one block, behind a carried half pair
    int have = 0;                                              // words waiting in w[0 .. have)
    while (i < n) {
        if (pos == kN) {
            regenerate(s, t);
            pos = 0;
        }
        const int64_t want = 2 * (n - i) - have;               // words still to take from the state
        const int take = (int)((int64_t)(kN - pos) < want ? (kN - pos) : want);
        for (int j = 0; j < take; ++j) w[have + j] = temper(s[pos + j]);
        pos += take;
        have += take;
        const int pairs = have / 2;
        if (flags && noise) kept += classify<true, true>(w, pairs, T, flags + i, noise + i, scale);
        else if (flags) kept += classify<true, false>(w, pairs, T, flags + i, nullptr, scale);
        else if (noise) kept += classify<false, true>(w, pairs, T, nullptr, noise + i, scale);
        else kept += classify<false, false>(w, pairs, T, nullptr, nullptr, scale);
        i += pairs;
        if (have & 1) w[0] = w[have - 1];
        have &= 1;
    }
    // `have` is 0 here (2 n words were taken in total); write the state back in torch's terms
    if (n > 0) {
        left = kN + 1 - pos;                                   // pos words of the current block consumed
        next = (uint64_t)pos;
        if (pos == kN) left = 1;                               // the next draw regenerates (left: 1 -> 0)
        seeded = 1;
        memcpy(rng_state + kOffLeft, &left, 4);
        memcpy(rng_state + kOffSeeded, &seeded, 4);
        memcpy(rng_state + kOffNext, &next, 8);
        for (int j = 0; j < kN; ++j) {
            const uint64_t wv = s[j];
            memcpy(rng_state + kOffState + 8 * (size_t)j, &wv, 8);
        }
    }
    if (n_kept) *n_kept = kept;
    return NGCF_OK;
}

// the draws of one forward, one after the other on the same state bytes (include/ngcf_hip.h): one call, so that a helper thread
// holds no interpreter lock from the first draw to the last
extern "C" int ngcf_torch_cpu_bernoulli_seq(uint8_t *rng_state, int64_t state_bytes, int n_draws, const int64_t *n, const double *keep,
                                            uint8_t *const *flags, float *const *noise, const float *scale, int64_t *n_kept)
{
    if (n_draws < 0 || (n_draws > 0 && (!n || !keep || !flags || !noise || !scale || !n_kept)))
        return fail(NGCF_ERR_ARG, "torch_cpu_bernoulli_seq: bad argument");
    int64_t last_kept = -1;
    for (int i = 0; i < n_draws; ++i) {
        const int64_t ni = n[i] >= 0 ? n[i] : last_kept;
        if (ni < 0) return fail(NGCF_ERR_ARG, "torch_cpu_bernoulli_seq: draw %d refers to an earlier draw with flags, there is none", i);
        const int rc = ngcf_torch_cpu_bernoulli(rng_state, state_bytes, ni, keep[i], flags[i], noise[i], scale[i], &n_kept[i]);
        if (rc != NGCF_OK) return rc;
        if (flags[i]) last_kept = n_kept[i];
    }
    return NGCF_OK;
}
