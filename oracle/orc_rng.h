// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// The reference draws every random number from rand 0.8.5's OS-seeded thread_rng()
// (21 call sites) and cannot be seeded. The restatement replaces the
// generator — and ONLY the generator — by a counter-based Philox4x32-10 stream so
// that a path is a pure function of (seed, pixel, sample); the ORDER of draws is the
// reference's.
//
//   key      = (seed_lo, pixel_index)
//   counter  = (draw_index >> 1, sample_index, seed_hi, 0)
//   draw 2k   -> u64 = out[1]<<32 | out[0];  draw 2k+1 -> u64 = out[3]<<32 | out[2]
//
// Distributions restate rand 0.8.5 (Cargo.lock:735-747; source not in the container,
// "parity unpinned"):
//   gen::<f64>()            = (u64 >> 11) * 2^-53                      in [0,1)
//   gen_range(0.0..=hi)     = ((u64 >> 12) * 2^-52) * scale, scale = hi / (1 - 2^-52)
//   gen_range(0..n) (usize) = widening multiply with rejection zone
#pragma once
#include <cstdint>

namespace orc {

struct Philox4 {
    uint32_t v[4];
};

inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                             uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return Philox4{{c0, c1, c2, c3}};
}

struct Rng {
    uint32_t seed_lo, seed_hi, pixel, sample;
    uint32_t draw = 0;       // number of u64 draws consumed so far for this sample
    uint64_t cache[2];
    uint32_t cached_block = 0xFFFFFFFFu;

    Rng(uint64_t seed, uint32_t pixel_, uint32_t sample_)
        : seed_lo((uint32_t)seed), seed_hi((uint32_t)(seed >> 32)), pixel(pixel_), sample(sample_) {}

    uint64_t next_u64() {
        uint32_t block = draw >> 1;
        if (block != cached_block) {
            Philox4 o = philox4x32_10(block, sample, seed_hi, 0u, seed_lo, pixel);
            cache[0] = ((uint64_t)o.v[1] << 32) | o.v[0];
            cache[1] = ((uint64_t)o.v[3] << 32) | o.v[2];
            cached_block = block;
        }
        uint64_t r = cache[draw & 1u];
        ++draw;
        return r;
    }
    // rand Standard f64
    double gen() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
    // rand UniformFloat<f64>::new_inclusive(0, hi).sample  (sampling.rs:20)
    double gen_range_inclusive(double hi) {
        const double max_rand = 1.0 - 1.0 / 4503599627370496.0;  // 1 - 2^-52
        double scale = hi / max_rand;
        while (!(scale * max_rand <= hi)) scale = std::nextafter(scale, 0.0);
        double v01 = (double)(next_u64() >> 12) * (1.0 / 4503599627370496.0);
        return v01 * scale;
    }
    // rand UniformInt<usize>::sample_single(0, n)  (list.rs:82)
    uint32_t gen_index(uint32_t n) {
        uint64_t range = n;
        int lz = __builtin_clzll(range);
        uint64_t zone = (range << lz) - 1;
        for (int it = 0; it < 64; ++it) {
            uint64_t v = next_u64();
            unsigned __int128 m = (unsigned __int128)v * range;
            uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
            if (lo <= zone) return (uint32_t)hi;
        }
        return 0;  // unreachable in practice (p = 2^-64)
    }
};

}  // namespace orc
