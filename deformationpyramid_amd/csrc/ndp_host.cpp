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

struct Mt {
    uint32_t s[N];
    int left;
    uint32_t next;
    inline void regen() {
        auto mix = [](uint32_t u, uint32_t v) { return (u & UMASK) | (v & LMASK); };
        auto tw = [&](uint32_t u, uint32_t v) { return (mix(u, v) >> 1) ^ ((v & 1u) ? MATRIX_A : 0u); };
        int j = 0;
        for (; j < N - M; ++j) s[j] = s[j + M] ^ tw(s[j], s[j + 1]);
        for (; j < N - 1; ++j) s[j] = s[j + M - N] ^ tw(s[j], s[j + 1]);
        s[N - 1] = s[M - 1] ^ tw(s[N - 1], s[0]);
        left = N;
        next = 0;
    }
    inline uint32_t raw() {              // at::mt19937::operator()
        if (--left == 0) regen();
        uint32_t y = s[next++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    inline void discard(long long n) {   // advance without tempering
        while (n > 0) {
            if (--left == 0) regen();
            // after the decrement `left` more words (this one included) are available: left+... take them in bulk
            long long avail = left;       // words still unread in this block, counting the current one
            long long take = n < avail ? n : avail;
            next += (uint32_t)take;
            left -= (int)(take - 1);
            n -= take;
        }
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
            float *dst = base + op.offset;
            const float lo = op.lo, span = op.hi - op.lo;
            for (long long i = 0; i < op.n; ++i) {
                const float x = (float)(g.raw() & ((1u << 24) - 1)) * (1.0f / 16777216.0f);
                dst[i] = __builtin_fmaf(x, span, lo);      // ATen's AVX2/AVX512 kernels contract x*(to-from)+from
            }
        }
    }
    export_state(g, rng_state);
    return 0;
}

int ndp_host_version(void) { return 100; }
}
