// CPU harness for tests/test_tune.py: the state machine that settles a scene's closest-hit switches (rt_tune.h) against
// synthetic timings -- what it runs on each call, that a faster flip is kept and a slower one dropped, that pins remove
// switches, that the walk ends after 1 + REPEAT (1 + number of switches) calls, that a flip inside the noise margin is
// not kept, and that a scene rendered only in calls too small to time gives up exploring.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../raytracer_amd/csrc/rt_tune.h"

using rt::GiTune;

#define CHECK(c)                                                      \
    do {                                                              \
        if (!(c)) {                                                   \
            std::printf("FAIL line %d: %s\n", __LINE__, #c);          \
            return 1;                                                 \
        }                                                             \
    } while (0)

// cost model of a made-up scene: ms per million rays as a function of the configuration
static double cost(uint32_t cfg, const double gain[4]) {
    double ms = 1.0;
    for (int b = 0; b < 4; ++b)
        if (cfg >> b & 1u) ms -= gain[b];
    return ms;
}

static uint32_t run(GiTune &t, const double gain[4], std::vector<uint32_t> *picks) {
    for (int call = 0; call < 16; ++call) {
        const uint32_t cfg = t.pick();
        picks->push_back(cfg);
        if (t.exploring()) t.record(cfg, cost(cfg, gain));
    }
    return t.pick();
}

int main() {
    {   // every switch open and helping: the starting configuration (two-phase GI queries, four waves) gains the other two;
        // 1 warm-up + 2 (1 + 4) timed calls
        GiTune t;
        bool ok[2] = {true, true};
        t.configure(ok, 3u, nullptr, nullptr);
        CHECK(t.n_dims == 4 && t.cfg == 3u && t.exploring());
        const double gain[4] = {0.1, 0.1, 0.1, 0.1};
        std::vector<uint32_t> picks;
        CHECK(run(t, gain, &picks) == 15u);
        CHECK(picks[0] == 3u && picks[1] == 3u && picks[2] == 3u);       // warm-up, then the base measurement (two calls)
        CHECK(picks[3] == 2u && picks[4] == 2u && picks[5] == 1u && picks[6] == 1u);  // one flip each, two calls: slower, dropped
        CHECK(picks[7] == 7u && picks[8] == 7u);                         // the last level's illumination nodes apart: faster, kept
        CHECK(picks[9] == 15u && picks[10] == 15u);                      // the build: faster, kept
        for (size_t i = 11; i < picks.size(); ++i) CHECK(picks[i] == 15u);
        CHECK(!t.exploring());
    }
    {   // two-phase on the inner levels hurts, three waves hurt: the first gets switched off, the others stay
        GiTune t;
        bool ok[2] = {true, true};
        t.configure(ok, 3u, nullptr, nullptr);
        const double gain[4] = {-0.2, 0.05, 0.04, -0.1};
        std::vector<uint32_t> picks;
        CHECK(run(t, gain, &picks) == (GiTune::GI_LAST | GiTune::ILLUM_LAST));
        CHECK(picks[3] == 2u);                  // GI_INNER off: faster, kept
        CHECK(picks[5] == (2u ^ 2u) && picks[7] == (2u ^ 4u) && picks[9] == (6u ^ 8u));
    }
    {   // a flip that is faster by less than the margin is noise: not kept (scene.json's [--N] against [---])
        GiTune t;
        bool ok[2] = {false, true};
        t.configure(ok, 3u, nullptr, nullptr);
        CHECK(t.n_dims == 3 && t.cfg == 2u);  // only the last level's set is small (physical.json)
        const double gain[4] = {0, -0.01, 0.02, -0.1};  // switching GI_LAST off / ILLUM_LAST on would gain 1 % / 2 %
        std::vector<uint32_t> picks;
        CHECK(run(t, gain, &picks) == 2u);
        CHECK(!t.exploring());
        GiTune far;  // a long occlusion distance (start = 0): the same switches are open, the walk starts with them off
        bool ok_far[2] = {true, true};
        far.configure(ok_far, 0u, nullptr, nullptr);
        CHECK(far.n_dims == 4 && far.cfg == 0u);
        GiTune big;  // a start bit the sets do not allow is dropped
        bool ok_big[2] = {false, true};
        big.configure(ok_big, 7u, nullptr, nullptr);
        CHECK(big.n_dims == 3 && big.cfg == 6u);
    }
    {   // a scene whose special sets are both large: only the build is open
        GiTune t;
        bool ok[2] = {false, false};
        t.configure(ok, 3u, nullptr, nullptr);
        CHECK(t.n_dims == 1 && t.cfg == 0u);
        const double gain[4] = {0, 0, 0, 0.3};
        std::vector<uint32_t> picks;
        CHECK(run(t, gain, &picks) == GiTune::WAVES3);
        CHECK(picks[0] == 0u && picks[1] == 0u && picks[2] == 0u && picks[3] == 8u && picks[4] == 8u && picks[5] == 8u);
    }
    {   // calls too small to time: exploring ends after MAX_SKIPPED of them, the starting configuration stays
        GiTune t;
        bool ok[2] = {true, true};
        t.configure(ok, 3u, nullptr, nullptr);
        for (int i = 0; i < GiTune::MAX_SKIPPED - 1; ++i) t.skip();
        CHECK(t.exploring());
        t.skip();
        CHECK(!t.exploring() && t.pick() == 3u);
    }
    {   // pins: no switch left, nothing explored, the sets are forced
        GiTune t;
        bool ok[2] = {false, false};
        t.configure(ok, 3u, "1", "4");
        CHECK(ok[0] && ok[1] && t.n_dims == 0 && t.cfg == 7u && !t.exploring());
        t.record(7u, 1.0);
        CHECK(t.pick() == 7u);
        GiTune u;
        bool ok2[2] = {true, true};
        u.configure(ok2, 3u, "0", "3");
        CHECK(!ok2[0] && !ok2[1] && u.n_dims == 0 && u.cfg == 8u);
        GiTune v;
        bool ok3[2] = {true, false};
        v.configure(ok3, 3u, nullptr, "3");
        CHECK(v.n_dims == 1 && v.cfg == (1u | 8u) && v.dims[0] == GiTune::GI_INNER);
        GiTune w;  // the whole configuration pinned (profiler runs)
        bool ok4[2] = {false, false};
        w.configure(ok4, 7u, nullptr, nullptr, "10");
        CHECK(w.n_dims == 0 && !w.exploring() && w.pick() == 10u && !ok4[0] && ok4[1]);
    }
    std::printf("OK\n");
    return 0;
}
