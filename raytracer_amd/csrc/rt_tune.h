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
//   WAVES3     k_trace_gi / k_trace_illum_last built for three waves per SIMD (168 VGPRs) instead of four (128)
// Which setting is fastest depends on the scene (how far max_occlusion_distance reaches, how much of it the
// special primitives fill, whether the tree fits the caches).  After an untimed first call the library times the
// closest-hit kernels of REPEAT render calls (each of at least a million GI rays) with its starting configuration,
// then flips the switches the scene allows one after the other, REPEAT calls each, and keeps a flip only when its
// mean is at least MARGIN faster than the best so far -- two configurations within noise of each other (scene.json:
// [--N] 0.1067 against [---] 0.1065 ms per million rays) must not decide by the noise, or every process would settle
// elsewhere and no profile would describe the run that was timed.  1 + REPEAT (1 + switches) calls, 11 at most; a
// scene that is only ever rendered in calls too small to time gives up after MAX_SKIPPED of them and keeps its
// starting configuration.  MI355RT_GI_TWO_PHASE=0|1 and MI355RT_GI_WAVES=3|4 pin switches.
struct GiTune {
    enum : uint32_t { GI_INNER = 1, GI_LAST = 2, ILLUM_LAST = 4, WAVES3 = 8 };
    static constexpr int REPEAT = 2;
    static constexpr double MARGIN = 0.03;
    static constexpr int MAX_SKIPPED = 8;
    uint32_t cfg = 0;        // best configuration so far
    double best_ms = 0.0;    // its closest-hit kernels' ms per million closest-hit rays
    uint32_t dims[4] = {0, 0, 0, 0};  // switches that may be flipped
    int n_dims = 0;
    int step = -1;           // -1: warm-up call pending; 0: measuring cfg; i >= 1: trying cfg ^ dims[i - 1]; > n_dims: settled
    int rep = 0;             // calls recorded for the current step
    double acc = 0.0;        // their sum
    int skipped = 0;         // calls that could not be used while exploring
    double tried_ms[5] = {0.0, 0.0, 0.0, 0.0, 0.0};  // for the MI355RT_TIMING report
    uint32_t tried_cfg[5] = {0, 0, 0, 0, 0};
    // set_ok[0 / 1]: the special set of the inner levels / of the last traced level is small enough for two-phase queries
    // (in / out: the pins override it -- "1" forces two phases whatever the sets' sizes); pin_two_phase / pin_waves: the
    // values of MI355RT_GI_TWO_PHASE / MI355RT_GI_WAVES or NULL.
    // The STARTING configuration is a rule from scene facts (`start`: GI_INNER | GI_LAST | ILLUM_LAST bits, masked here by
    // what the sets allow; rt_api.hip scene_create_impl derives it), so that a process that renders one frame (the
    // reference's use, main.rs:58-77) already runs what the walk would settle on; the four-wave build (faster on all eight
    // published scenes since the traversal stopped spilling).
    // pin_cfg: MI355RT_GI_CFG = the whole configuration as a number (GI_INNER 1 | GI_LAST 2 | ILLUM_LAST 4 | WAVES3 8), for
    // profiler runs that must describe one variant: nothing is explored, the sets the bits need are forced usable.
    void configure(bool set_ok[2], uint32_t start, const char *pin_two_phase, const char *pin_waves, const char *pin_cfg = nullptr) {
        cfg = 0;
        n_dims = 0;
        step = -1;
        rep = skipped = 0;
        acc = 0.0;
        if (pin_cfg) {
            uint32_t v = 0;
            for (const char *c = pin_cfg; *c >= '0' && *c <= '9'; ++c) v = v * 10u + static_cast<uint32_t>(*c - '0');
            cfg = v & 15u;
            if (cfg & GI_INNER) set_ok[0] = true;
            if (cfg & (GI_LAST | ILLUM_LAST)) set_ok[1] = true;
            return;
        }
        if (pin_two_phase) {
            set_ok[0] = set_ok[1] = std::strcmp(pin_two_phase, "0") != 0;
            if (set_ok[0]) cfg = GI_INNER | GI_LAST | ILLUM_LAST;
        } else {
            if (set_ok[0]) cfg |= start & GI_INNER, dims[n_dims++] = GI_INNER;
            if (set_ok[1]) cfg |= start & (GI_LAST | ILLUM_LAST), dims[n_dims++] = GI_LAST, dims[n_dims++] = ILLUM_LAST;
        }
        if (pin_waves) {
            if (std::strcmp(pin_waves, "3") == 0) cfg |= WAVES3;
        } else {
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
        acc += ms;
        if (++rep < REPEAT) return;
        const double mean = acc / REPEAT;
        acc = 0.0;
        rep = 0;
        tried_cfg[step] = ran;
        tried_ms[step] = mean;
        if (step == 0 || mean < best_ms * (1.0 - MARGIN)) {
            cfg = ran;
            best_ms = mean;
        }
        ++step;
    }
    void skip() {  // a call that could not be timed (too small, re-run batches, work counting)
        if (exploring() && ++skipped >= MAX_SKIPPED) step = n_dims + 1;  // keep what is best so far
    }
};

}  // namespace rt
