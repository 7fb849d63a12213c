// CPU harness for tests/test_tune.py: the state machine that settles a scene's closest-hit switches (rt_tune.h) against
// synthetic timings -- what it runs on each call, that a faster flip is kept and a slower one dropped, that pins remove
// switches, and that the walk ends after 1 + (number of switches) timed calls.
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
    for (int call = 0; call < 12; ++call) {
        const uint32_t cfg = t.pick();
        picks->push_back(cfg);
        if (t.exploring()) t.record(cfg, cost(cfg, gain));
    }
    return t.pick();
}

int main() {
    {   // every switch open, all of them help: the starting configuration stays; 1 warm-up + 1 + 4 timed calls
        GiTune t;
        bool ok[2] = {true, true};
        t.configure(ok, false, nullptr, nullptr);
        CHECK(t.n_dims == 4 && t.cfg == 15u && t.exploring());
        const double gain[4] = {0.1, 0.1, 0.1, 0.1};
        std::vector<uint32_t> picks;
        CHECK(run(t, gain, &picks) == 15u);
        CHECK(picks[0] == 15u && picks[1] == 15u);                        // warm-up, then the base measurement
        CHECK(picks[2] == 14u && picks[3] == 13u && picks[4] == 11u && picks[5] == 7u);  // one flip each
        for (size_t i = 6; i < picks.size(); ++i) CHECK(picks[i] == 15u);
        CHECK(!t.exploring());
    }
    {   // two-phase on the inner levels hurts, three waves hurt: both get switched off, the others stay
        GiTune t;
        bool ok[2] = {true, true};
        t.configure(ok, false, nullptr, nullptr);
        const double gain[4] = {-0.2, 0.05, 0.02, -0.1};
        std::vector<uint32_t> picks;
        CHECK(run(t, gain, &picks) == (GiTune::GI_LAST | GiTune::ILLUM_LAST));
        CHECK(picks[2] == 14u);                  // GI_INNER off: faster, kept
        CHECK(picks[3] == (14u ^ 2u) && picks[4] == (14u ^ 4u) && picks[5] == (14u ^ 8u));
    }
    {   // a scene whose special sets are both large: only the build is open; a large scene starts from four waves
        GiTune t;
        bool ok[2] = {false, false};
        t.configure(ok, true, nullptr, nullptr);
        CHECK(t.n_dims == 1 && t.cfg == 0u);
        const double gain[4] = {0, 0, 0, 0.3};
        std::vector<uint32_t> picks;
        CHECK(run(t, gain, &picks) == GiTune::WAVES3);
        CHECK(picks[0] == 0u && picks[1] == 0u && picks[2] == 8u && picks[3] == 8u);
    }
    {   // only the last level's set is small (physical.json)
        GiTune t;
        bool ok[2] = {false, true};
        t.configure(ok, false, nullptr, nullptr);
        CHECK(t.n_dims == 3 && t.cfg == (2u | 4u | 8u));
    }
    {   // pins: no switch left, nothing explored, the sets are forced
        GiTune t;
        bool ok[2] = {false, false};
        t.configure(ok, false, "1", "4");
        CHECK(ok[0] && ok[1] && t.n_dims == 0 && t.cfg == 7u && !t.exploring());
        t.record(7u, 1.0);
        CHECK(t.pick() == 7u);
        GiTune u;
        bool ok2[2] = {true, true};
        u.configure(ok2, false, "0", "3");
        CHECK(!ok2[0] && !ok2[1] && u.n_dims == 0 && u.cfg == 8u);
        GiTune v;
        bool ok3[2] = {true, false};
        v.configure(ok3, false, nullptr, "3");
        CHECK(v.n_dims == 1 && v.cfg == (1u | 8u) && v.dims[0] == GiTune::GI_INNER);
    }
    std::printf("OK\n");
    return 0;
}
