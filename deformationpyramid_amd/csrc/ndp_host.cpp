// ndp_host.cpp -- host-side runtime helpers (plain C++, no GPU): exact replay of the torch CPU
// generator for the NDP parameter initialisation.
//
// Why: Deformation_Pyramid.__init__ (/root/reference/model/nets.py:10-34,180-183) draws ~620 000
// uniform numbers per pair from torch's global mt19937 (default nn.Linear init, then xavier_uniform_
// on every matrix).  Done through torch.Tensor.uniform_ this costs ~3 ms per pair and caps a GPU that
// registers a pair in ~2 ms of device time.  Here the same stream is produced directly:
//   * the engine is torch's at::mt19937 (standard MT19937, 624-word block regeneration, state imported
//     from / exported to torch.get_rng_state() so torch.randperm etc. continue seamlessly);
//   * uniform_(from, to) on a float tensor is at::uniform_real_distribution<float>:
//         x = (random() & (2^24 - 1)) * 2^-24 ;  value = x * (to - from) + from        (float arithmetic)
//   * draws whose values the reference overwrites (the kaiming init of every weight matrix, replaced by
//     xavier_uniform_ a moment later) only advance the generator.
// Bit-exactness against torch is asserted by tests/test_oracle_golden.py::test_F1* (golden F1 comes
// from the reference itself) and tests/test_host_cpu.py::test_native_rng_replay_matches_torch.
#include <cstdint>
#include <cstring>

namespace {
constexpr int N = 624, M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UMASK = 0x80000000u, LMASK = 0x7fffffffu;

inline uint32_t temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// at::mt19937 with torch's (left, next) bookkeeping: operator() is `if (--left == 0) regen(); return temper(s[next++]);`
// -- with `left` = L on entry the next L - 1 calls read s[next ..] without a regeneration, the L-th regenerates first.
struct Mt {
    uint32_t s[N];
    int left;
    uint32_t next;
    inline void regen() {
        // branch-free twist; neither loop carries a dependency closer than 227 words, so both vectorise (-mavx2)
        auto tw = [](uint32_t u, uint32_t v) { return (((u & UMASK) | (v & LMASK)) >> 1) ^ ((0u - (v & 1u)) & MATRIX_A); };
        int j = 0;
        for (; j < N - M; ++j) s[j] = s[j + M] ^ tw(s[j], s[j + 1]);
        for (; j < N - 1; ++j) s[j] = s[j + M - N] ^ tw(s[j], s[j + 1]);
        s[N - 1] = s[M - 1] ^ tw(s[N - 1], s[0]);
        left = N;
        next = 0;
    }
    inline uint32_t raw() {              // at::mt19937::operator()
        if (--left == 0) regen();
        return temper(s[next++]);
    }
    // n consecutive draws handed to f(pointer to untempered words, count) in bulk -- same stream as n calls of raw()
    template <class F>
    inline void bulk(long long n, F &&f) {
        while (n > 0) {
            if (left == 1) {             // the next call regenerates, reads s[0] and leaves left = N
                regen();
                f(s, 1LL);
                next = 1;
                --n;
                continue;
            }
            const long long take = n < (long long)(left - 1) ? n : (long long)(left - 1);
            f(s + next, take);
            next += (uint32_t)take;
            left -= (int)take;
            n -= take;
        }
    }
    inline void discard(long long n) {   // advance without tempering
        bulk(n, [](const uint32_t *, long long) {});
    }
    // n values ~ U(lo, lo + span) exactly as at::uniform_real_distribution<float> on torch's CPU generator
    inline void uniform(float *dst, long long n, float lo, float span) {
        bulk(n, [&](const uint32_t *w, long long k) {
            for (long long i = 0; i < k; ++i) {
                const float x = (float)(temper(w[i]) & ((1u << 24) - 1)) * (1.0f / 16777216.0f);
                dst[i] = __builtin_fmaf(x, span, lo);      // ATen's AVX2/AVX512 kernels contract x*(to-from)+from
            }
            dst += k;
        });
    }
};

// torch.get_rng_state() layout (CPUGeneratorImplStateLegacy): u64 seed, i32 left, i32 seeded, u64 next, u64 state[624], ...
inline void import_state(const unsigned char *st, Mt &g) {
    int32_t left;
    uint64_t next;
    std::memcpy(&left, st + 8, 4);
    std::memcpy(&next, st + 16, 8);
    const unsigned char *p = st + 24;
    for (int i = 0; i < N; ++i) { uint64_t v; std::memcpy(&v, p + 8 * i, 8); g.s[i] = (uint32_t)v; }
    g.left = left;
    g.next = (uint32_t)next;
}
inline void export_state(const Mt &g, unsigned char *st) {
    int32_t left = g.left;
    uint64_t next = g.next;
    std::memcpy(st + 8, &left, 4);
    std::memcpy(st + 16, &next, 8);
    unsigned char *p = st + 24;
    for (int i = 0; i < N; ++i) { uint64_t v = g.s[i]; std::memcpy(p + 8 * i, &v, 8); }
}
}  // namespace

extern "C" {

// One draw op: n values ~ U(lo, hi) written to out + offset (offset < 0: discard n draws).
struct ndp_draw_op { long long n; float lo, hi; long long offset; };

// Executes `n_ops` ops, `repeat` times, on the generator state `rng_state` (the 5056-byte blob of
// torch.get_rng_state(), updated in place).  Repetition r writes at out + r * out_stride.
// Returns 0, or -1 on bad arguments.
int ndp_rng_replay(unsigned char *rng_state, long long state_bytes, const ndp_draw_op *ops, int n_ops,
                   int repeat, float *out, long long out_stride) {
    if (!rng_state || state_bytes < 24 + 8 * N || !ops || n_ops < 0 || repeat < 0) return -1;
    Mt g;
    import_state(rng_state, g);
    for (int r = 0; r < repeat; ++r) {
        float *base = out + (long long)r * out_stride;
        for (int k = 0; k < n_ops; ++k) {
            const ndp_draw_op &op = ops[k];
            if (op.offset < 0) { g.discard(op.n); continue; }
            g.uniform(base + op.offset, op.n, op.lo, op.hi - op.lo);
        }
    }
    export_state(g, rng_state);
    return 0;
}

// ---- multi-threaded pair producer (registration.py:133-159 for MANY pairs) ---------------------------------------
// The per-pair draw counts are known up front (the init ops + the two randperm calls), so ONE thread can walk the
// generator from pair to pair (ndp_rng_skip: regenerations only, no tempering / float transform) and hand every pair
// the generator state it starts from; any number of workers then replay their pair from that snapshot
// (ndp_pair_init) -- bit-identical to the sequential order of Registration.register() calls.

// advance the generator state blob by n raw draws
int ndp_rng_skip(unsigned char *rng_state, long long state_bytes, long long n) {
    if (!rng_state || state_bytes < 24 + 8 * N || n < 0) return -1;
    Mt g;
    import_state(rng_state, g);
    g.discard(n);
    export_state(g, rng_state);
    return 0;
}

// first `keep` entries of torch.randperm(n) on the CPU generator (randperm_cpu, n < 2^32 / 20: r = arange(n); for i < n - 1:
// swap(r[i], r[i + random() % (n - i)]) -- entry i is final after step i); always consumes n - 1 draws.  scratch: n ints.
static void randperm_prefix(Mt &g, int n, int keep, int *scratch, int *out) {
    if (n <= 0) return;
    for (int i = 0; i < n; ++i) scratch[i] = i;
    const int steps = keep < n - 1 ? keep : n - 1;
    for (int i = 0; i < steps; ++i) {
        const int z = (int)(g.raw() % (uint32_t)(n - i));
        const int a = scratch[i];
        scratch[i] = scratch[z + i];
        scratch[z + i] = a;
    }
    g.discard((long long)(n - 1) - steps);
    const int k = keep < n ? keep : n;
    for (int i = 0; i < k; ++i) out[i] = scratch[i];
}

// raw draws one pair consumes: the init ops + randperm(n_src) + randperm(n_tgt)
long long ndp_pair_draws(const ndp_draw_op *ops, int n_ops, int n_src, int n_tgt) {
    long long tot = 0;
    for (int k = 0; k < n_ops; ++k) tot += ops[k].n;
    return tot + (n_src > 1 ? n_src - 1 : 0) + (n_tgt > 1 ? n_tgt - 1 : 0);
}

// One pair from a state SNAPSHOT (read only): the pyramid initialisation into `out` (draw ops as ndp_rng_replay) and the first
// `samples` entries of the two sampling permutations (registration.py:156-159) into perm_s / perm_t (int32).
// scratch: max(n_src, n_tgt) ints.  Thread safe (no shared state).
int ndp_pair_init(const unsigned char *rng_state, long long state_bytes, const ndp_draw_op *ops, int n_ops, float *out,
                  int n_src, int n_tgt, int samples, int *perm_s, int *perm_t, int *scratch) {
    if (!rng_state || state_bytes < 24 + 8 * N || !ops || n_ops < 0 || !out || n_src < 1 || n_tgt < 1 || samples < 0 || !perm_s ||
        !perm_t || !scratch)
        return -1;
    Mt g;
    import_state(rng_state, g);
    for (int k = 0; k < n_ops; ++k) {
        const ndp_draw_op &op = ops[k];
        if (op.offset < 0) g.discard(op.n);
        else g.uniform(out + op.offset, op.n, op.lo, op.hi - op.lo);
    }
    randperm_prefix(g, n_src, samples, scratch, perm_s);
    randperm_prefix(g, n_tgt, samples, scratch, perm_t);
    return 0;
}

#ifndef NDP_HOST_BUILD_ID
#define NDP_HOST_BUILD_ID "unknown"
#endif
static const char k_host_tag[] = "NDP_HOST_ID=" NDP_HOST_BUILD_ID;       // the loader finds this tag in the file without loading it
const char *ndp_host_build_id(void) { return k_host_tag + 12; }
int ndp_host_version(void) { return 110; }
}
