// Which variant of the closest-hit kernels a scene uses, settled by measurement on its first frames (host logic only:
// no HIP in here, so tests/native/tune_check.cpp can drive it on the CPU).
#pragma once
#include <cstdint>
#include <cstring>

namespace rt {

// The closest-hit work of a frame has switches that leave the frame the same bit for bit and change its speed:
//   GI_INNER   k_trace_gi answers the GI rays of the inner levels with two-phase queries (rt_kernels.hip.h)
//   GI_LAST    ... and those of the last traced level (there only emissive primitives are special)
//   ILLUM_LAST the last level's illumination NODES go to k_trace_illum_last (two-phase) instead of k_trace_closest
//   WAVES3     k_trace_gi / k_trace_illum_last built for three waves per SIMD (168 VGPRs, no spills) instead of four
// Which setting is fastest depends on the scene (how far max_occlusion_distance reaches, how much of it the
// special primitives fill, whether the tree fits the caches).  After an untimed first call the library times the
// closest-hit kernels of one render call (of at least a million GI rays) with its starting configuration, then
// flips the switches the scene allows one after the other, a call each, keeping a flip that is faster: 1 + (number
// of switches) calls, 5 at most.  MI355RT_GI_TWO_PHASE=0|1 and MI355RT_GI_WAVES=3|4 pin switches.
struct GiTune {
    enum : uint32_t { GI_INNER = 1, GI_LAST = 2, ILLUM_LAST = 4, WAVES3 = 8 };
    uint32_t cfg = 0;        // best configuration so far
    double best_ms = 0.0;    // its closest-hit kernels' ms per million closest-hit rays
    uint32_t dims[4] = {0, 0, 0, 0};  // switches that may be flipped
    int n_dims = 0;
    int step = -1;           // -1: warm-up call pending; 0: measuring cfg; i >= 1: trying cfg ^ dims[i - 1]; > n_dims: settled
    double tried_ms[5] = {0.0, 0.0, 0.0, 0.0, 0.0};  // for the MI355RT_TIMING report
    uint32_t tried_cfg[5] = {0, 0, 0, 0, 0};
    // set_ok[0 / 1]: the special set of the inner levels / of the last traced level is small enough for two-phase queries
    // (in / out: the pins override it -- "1" forces two phases whatever the sets' sizes); large_scene: start from the
    // 4-wave build; pin_two_phase / pin_waves: the values of MI355RT_GI_TWO_PHASE / MI355RT_GI_WAVES or NULL.
    void configure(bool set_ok[2], bool large_scene, const char *pin_two_phase, const char *pin_waves) {
        cfg = 0;
        n_dims = 0;
        step = -1;
        if (pin_two_phase) {
            set_ok[0] = set_ok[1] = std::strcmp(pin_two_phase, "0") != 0;
            if (set_ok[0]) cfg = GI_INNER | GI_LAST | ILLUM_LAST;
        } else {
            if (set_ok[0]) cfg |= GI_INNER, dims[n_dims++] = GI_INNER;
            if (set_ok[1]) cfg |= GI_LAST | ILLUM_LAST, dims[n_dims++] = GI_LAST, dims[n_dims++] = ILLUM_LAST;
        }
        if (pin_waves) {
            if (std::strcmp(pin_waves, "3") == 0) cfg |= WAVES3;
        } else {
            if (!large_scene) cfg |= WAVES3;
            dims[n_dims++] = WAVES3;
        }
    }
    bool exploring() const { return n_dims > 0 && step <= n_dims; }
    uint32_t pick() const { return step >= 1 && step <= n_dims ? cfg ^ dims[step - 1] : cfg; }
    void record(uint32_t ran, double ms) {  // a call with configuration `ran` measured `ms`
        if (step < 0) {
            step = 0;  // the scene's first sizeable call is not used (cold caches and clocks, pool sizing)
            return;
        }
        if (step > n_dims) return;
        tried_cfg[step] = ran;
        tried_ms[step] = ms;
        if (step == 0 || ms < best_ms) {
            cfg = ran;
            best_ms = ms;
        }
        ++step;
    }
};

}  // namespace rt
