// host/pcg64.hpp - the reference's seeded scene-build generator, host side.
//
// SeedableRngator::rng(site) = rand_pcg::Pcg64::seed_from_u64(seed + site)   (/root/reference/src/rngator.rs:27-31)
// rand_pcg 0.3.0 Pcg64 = Lcg128Xsl64, rand_core 0.6.2 seed_from_u64, rand 0.8.3 gen_range (Cargo.lock pins).
// The device path uses Philox for rendering; this generator only replays the world-build stream so that the
// scene at a given --seed is the reference's scene (main.rs:185-188).
#pragma once
#include <cstdint>
#include <cstring>

namespace mlrt {

class Pcg64 {
  public:
    typedef unsigned __int128 u128;
    static Pcg64 seed_from_u64(uint64_t seed) {
        // rand_core: expand the u64 with PCG32 into 32 seed bytes (eight little-endian u32 words)
        uint32_t w[8];
        for (uint32_t& out : w) {
            seed = seed * 6364136223846793005ULL + 11634580027462260723ULL;
            const uint32_t xs = static_cast<uint32_t>(((seed >> 18) ^ seed) >> 27);
            const uint32_t rot = static_cast<uint32_t>(seed >> 59);
            out = rot ? ((xs >> rot) | (xs << (32 - rot))) : xs;
        }
        auto pair = [&](int i) { return static_cast<uint64_t>(w[i]) | (static_cast<uint64_t>(w[i + 1]) << 32); };
        Pcg64 g;
        // Lcg128Xsl64::from_seed -> from_state_incr(state, incr | 1): state += incr; step
        g.state_ = (static_cast<u128>(pair(2)) << 64) | pair(0);
        g.inc_ = ((static_cast<u128>(pair(6)) << 64) | pair(4)) | 1;
        g.state_ += g.inc_;
        g.advance();
        return g;
    }
    uint64_t next_u64() {
        advance();
        const unsigned rot = static_cast<unsigned>(state_ >> 122);
        const uint64_t x = static_cast<uint64_t>(state_ >> 64) ^ static_cast<uint64_t>(state_);
        return rot ? ((x >> rot) | (x << (64 - rot))) : x;
    }
    // rand 0.8.3 UniformFloat<f64>::sample_single: 52 random mantissa bits in [1,2), minus 1, times scale, plus low
    double gen_range(double low, double high) {
        const double scale = high - low;
        for (;;) {
            const uint64_t bits = (next_u64() >> 12) | 0x3FF0000000000000ULL;
            double one_two;
            std::memcpy(&one_two, &bits, sizeof one_two);
            const double r = (one_two - 1.0) * scale + low;
            if (r < high) return r;
        }
    }
    // rand 0.8.3 UniformInt<usize>::sample_single on a 64-bit target (bhv.rs:127, textures.rs:142)
    uint64_t gen_range_usize(uint64_t low, uint64_t high) {
        const uint64_t range = high - low;
        const uint64_t zone = (range << __builtin_clzll(range)) - 1;
        for (;;) {
            const u128 wide = static_cast<u128>(next_u64()) * range;
            if (static_cast<uint64_t>(wide) <= zone) return low + static_cast<uint64_t>(wide >> 64);
        }
    }

  private:
    void advance() {
        const u128 mul = (static_cast<u128>(0x2360ED051FC65DA4ULL) << 64) | 0x4385DF649FCCF645ULL;
        state_ = state_ * mul + inc_;
    }
    u128 state_ = 0, inc_ = 1;
};

}  // namespace mlrt
