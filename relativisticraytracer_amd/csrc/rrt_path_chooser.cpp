// rrt_path_chooser.cpp -- which path a rank's share takes while several frames are in flight (host only, g++).
//
// A launch of a rank's share can take the three-pass path (march_defer -> eval_sample_rows -> composite_and_shade:
// RRT_PATH_AUTO picks it for <= 1.5 M rays when a pool is given) or the plain single kernel (RRT_PATH_SINGLE).  With other
// frames of the sequence in flight the single kernel is 3-10 % faster -- the other frames fill its drain, it does 4 % less
// work and has no pool traffic -- EXCEPT where the share holds a wavefront that outlasts the frames in flight together
// (a disk-grazing view: one wavefront of 19 ms): a slot's next frame waits for it, and the three-pass path, whose longest
// wave only marches, is 20 % faster (profiles/r05_sustained_chains.txt).  Which case a rank is in depends on the view, and
// the animation drivers' camera moves: so each rank measures.  bench.py (fixed camera) does it once at start-up; the two
// headless drivers ask this object per frame.
//
// Rule (per rank; the bytes do not depend on it).  S = frames in flight, T = 4 S (a trial), C = T - S (its counted frames):
//   - the sequence is cut into windows of `window_frames` frames.  The INCUMBENT path renders a window, except for a TRIAL of
//     the other path over its first T frames.  Frames whose neighbours in flight ran the other path are not counted: the first
//     S frames of the trial and the S frames after it;
//   - what is reported is a frame's interval on the rank: the time between the ends of consecutive frames' renders (with
//     frames in flight a frame's own start-to-end latency says nothing).  With S frames in flight on S streams these intervals
//     come in a pattern of period S (two frames end together, then a gap: measured, profiles/r06_chooser_probe.txt -- the
//     first version compared medians and took every other gap for an outlier), so what is COMPARED are means over C = 3 S
//     frames = elapsed time per frame, i.e. the sustained throughput of a path;
//   - the camera MOVES, so a trial is compared with the incumbent's frames on BOTH sides of it: block A = the last C counted
//     incumbent frames before the trial, block B = the first C after it; the incumbent's figure is (mean A + mean B) / 2, which
//     cancels a workload that drifts linearly across the 2-3 T frames involved (the second version compared the trial with the
//     REST of the window: on the reference's moving paths it flapped and lost 5 %: LABNOTES round 6);
//   - the decision is taken as soon as block B is in: the other path becomes the incumbent, from the next frame on, if it wins
//     by 4 % (hysteresis: the mean of nine frames of a rank's share is good to about 2 %).  A single kernel that did not win is
//     tried again after 2, then 4 windows; an incumbent single kernel is checked every second window;
//   - the first window is short (T + 1 frames), three-pass only, no trial: it supplies the first block A and the reference of
//     the outlier rule;
//   - S consecutive single-kernel frames that together take more than `outlier` (2.5) x as long as the reference (the three-pass
//     path's last mean; for an incumbent single kernel also its own recent mean: an outlier is a jump, not the drift of a moving
//     camera -- at 1.5 x of a stale three-pass mean the rule demoted a healthy single kernel on the Horizon Skimmer path) end a
//     trial of the single kernel at once -- or, if the single kernel is the incumbent, hand the rest of the window to the
//     three-pass path -- and double the distance to the next trial (up to every 8th window).  It is a guard against a share
//     that suddenly holds a wavefront longer than the frames in flight together; the ordinary case of a slower path is decided
//     by the means;
//   - a three-pass trial that loses by more than 10 % makes the next one rarer too (every 4th, then 8th window): its 12 frames
//     are the expensive ones where the single kernel is far ahead.
// The reports arrive late (a frame's interval is known once it has been delivered, S frames after it was enqueued).
#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/rrt.h"

namespace {

struct Chooser {
    int slots = 3, window = 48, trial = 12, counted = 9;
    float outlier = 2.5f, hysteresis = 0.96f;
    int incumbent = RRT_PATH_AUTO;
    int win_start = 1;               // first frame of the current window
    int win_index = 0;               // windows begun
    int first_window = 13;           // frames of the first window
    int trial_every = 1;             // incumbent three-pass: a trial of the single kernel at the start of every `trial_every`-th window
    int since_trial = 0;
    bool trial_on = false;           // this window has (had) a trial
    bool trial_aborted = false, decided = false;
    int trial_end = 0;               // first frame after the trial
    int trial_policy = RRT_PATH_SINGLE;
    bool demoted = false;            // the incumbent single kernel lost the rest of this window to the three-pass path
    int demoted_from = 0;
    int clean_from = 1;              // incumbent frames from here on have only incumbent neighbours in flight
    std::vector<int8_t> policy_of;   // policy_of[frame % size]
    std::vector<float> tail;         // the last `counted` clean incumbent intervals (block A of the next trial)
    std::vector<float> block_a, trial_ms, block_b;
    float recent_ms[16] = {};        // the last `slots` reports, if they were consecutive single-kernel frames (the outlier rule)
    int recent_n = 0, recent_last = 0;
    rrt_path_chooser_stats st = {};
};

std::mutex g_mu;
std::vector<Chooser*> g_choosers;    // id - 1 -> object (nullptr once destroyed)

Chooser* get(int id) {
    if (id < 1 || (size_t)id > g_choosers.size()) return nullptr;
    return g_choosers[(size_t)id - 1];
}

float mean(const std::vector<float>& v) {
    if (v.empty()) return 0.0f;
    double s = 0.0;
    for (float x : v) s += x;
    return (float)(s / (double)v.size());
}

void begin_window(Chooser& c, int frame) {
    c.win_start = frame;
    ++c.win_index;
    // the first window renders with the three-pass path only.  With the single kernel as incumbent every SECOND window has its
    // (three-pass) trial; trials OF the single kernel back off after one it did not win or that was aborted.
    c.trial_on = c.win_index > 1 && ++c.since_trial >= (c.incumbent == RRT_PATH_SINGLE ? std::max(2, c.trial_every) : c.trial_every);
    if (c.trial_on) c.since_trial = 0;
    c.trial_aborted = false;
    c.decided = false;
    c.trial_policy = c.incumbent == RRT_PATH_SINGLE ? RRT_PATH_AUTO : RRT_PATH_SINGLE;
    c.trial_end = frame + (c.trial_on ? c.trial : 0);
    if (c.demoted) {                                  // the single kernel met a wavefront that outlasts the frames in flight
        c.incumbent = RRT_PATH_AUTO;
        ++c.st.switches;
        c.demoted = false;
        c.tail.clear();
        c.trial_on = false;
        c.trial_end = frame;
        c.clean_from = frame + c.slots;
    }
    c.recent_n = 0;
    c.block_a = c.tail;
    c.trial_ms.clear(); c.block_b.clear();
    if (c.trial_on) { ++c.st.trials; c.clean_from = c.trial_end + c.slots; }
}

// the trial against the incumbent's frames on both sides of it
void decide(Chooser& c) {
    c.decided = true;
    if (c.trial_aborted || c.trial_ms.size() < (size_t)c.slots || c.block_b.size() < (size_t)c.slots) return;
    const float alt = mean(c.trial_ms);
    const float inc = c.block_a.size() >= (size_t)c.slots ? 0.5f * (mean(c.block_a) + mean(c.block_b)) : mean(c.block_b);
    if (alt < c.hysteresis * inc) {
        c.incumbent = c.trial_policy;
        ++c.st.switches;
        c.trial_every = 1;
        c.tail.clear();                               // intervals of the old incumbent say nothing about the new one
    } else if (c.incumbent == RRT_PATH_AUTO) {
        c.trial_every = std::min(4, c.trial_every * 2);       // the single kernel did not win: look again in two windows, then in four
    } else if (alt > 1.10f * inc) {
        c.trial_every = std::min(8, std::max(2, c.trial_every) * 2);    // the three-pass path lost by > 10 %: its trials (12 slow frames) get rarer
    } else {
        c.trial_every = 2;
    }
}

}  // namespace

extern "C" {

int rrt_path_chooser_create(int frames_in_flight, int window_frames, int* out_id) {
    if (!out_id || frames_in_flight < 1 || frames_in_flight > 16) return RRT_ERR_INVALID_ARGUMENT;
    Chooser* c = new (std::nothrow) Chooser();
    if (!c) return RRT_ERR_OUT_OF_MEMORY;
    c->slots = frames_in_flight;
    c->trial = 4 * frames_in_flight;
    c->counted = c->trial - frames_in_flight;
    c->window = window_frames > 0 ? window_frames : 48;
    const int least = c->trial + c->slots + 2 * c->counted;        // trial, its wake, block B, and a block A for the next window
    if (c->window < least) c->window = least;
    c->first_window = c->trial + 1;
    c->policy_of.assign(1024, (int8_t)RRT_PATH_AUTO);
    std::lock_guard<std::mutex> lk(g_mu);
    g_choosers.push_back(c);
    *out_id = (int)g_choosers.size();
    return RRT_OK;
}

int rrt_path_chooser_destroy(int id) {
    std::lock_guard<std::mutex> lk(g_mu);
    Chooser* c = get(id);
    if (!c) return RRT_ERR_BAD_HANDLE;
    delete c;
    g_choosers[(size_t)id - 1] = nullptr;
    return RRT_OK;
}

/* the path_policy of frame `frame` (1-based; frames are asked for in increasing order) */
int rrt_path_chooser_policy(int id, int frame, int* policy_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    Chooser* c = get(id);
    if (!c) return RRT_ERR_BAD_HANDLE;
    if (!policy_out || frame < 1) return RRT_ERR_INVALID_ARGUMENT;
    if (c->win_index == 0) begin_window(*c, frame);
    else if (frame >= c->win_start + (c->win_index == 1 ? c->first_window : c->window)) {
        if (c->trial_on && !c->decided) decide(*c);              // the reports of block B came late: decide on what is there
        begin_window(*c, frame);
    }
    int p = c->incumbent;
    if (c->trial_on && !c->trial_aborted && frame < c->trial_end) p = c->trial_policy;
    if (c->demoted && frame >= c->demoted_from) p = RRT_PATH_AUTO;
    c->policy_of[(size_t)frame % c->policy_of.size()] = (int8_t)p;
    ++c->st.frames[p == RRT_PATH_SINGLE ? 1 : 0];
    *policy_out = p;
    return RRT_OK;
}

/* the interval of frame `frame` on this rank: milliseconds between the end of frame `frame - 1`'s render and its own */
int rrt_path_chooser_report(int id, int frame, float ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    Chooser* c = get(id);
    if (!c) return RRT_ERR_BAD_HANDLE;
    if (frame < 1 || !(ms >= 0.0f)) return RRT_ERR_INVALID_ARGUMENT;
    if (c->win_index == 0 || frame < c->win_start) return RRT_OK;              // a report of an earlier window: too late to matter
    const int p = c->policy_of[(size_t)frame % c->policy_of.size()];
    const bool in_trial = c->trial_on && frame < c->trial_end;
    const bool trial_counted = in_trial && frame >= c->win_start + c->slots && p == c->trial_policy;
    const bool clean_incumbent = !in_trial && frame >= c->clean_from && p == c->incumbent && !(c->demoted && frame >= c->demoted_from);
    // ---- the outlier rule on `slots` consecutive clean single-kernel frames together (one period of the interval pattern)
    if (p == RRT_PATH_SINGLE && (trial_counted || clean_incumbent)) {
        if (c->recent_n > 0 && frame != c->recent_last + 1) c->recent_n = 0;
        if (c->recent_n == c->slots) { for (int i = 1; i < c->slots; ++i) c->recent_ms[i - 1] = c->recent_ms[i]; --c->recent_n; }
        c->recent_ms[c->recent_n++] = ms;
        c->recent_last = frame;
        // reference: the three-pass path's last mean, or -- the workload of an animation drifts, and a mean measured a window ago
        // may be half of today's -- the single kernel's own recent mean if that is larger: an outlier is a JUMP, not a drift
        float ref = c->st.last_three_pass_mean_ms;
        if (c->incumbent == RRT_PATH_SINGLE && c->tail.size() >= (size_t)(2 * c->slots)) {
            // (the tail WITHOUT its last `slots` entries: those may be the very frames under test)
            const std::vector<float> older(c->tail.begin(), c->tail.end() - c->slots);
            ref = std::max(ref, mean(older));
        }
        if (c->recent_n == c->slots && ref > 0.0f) {
            float sum = 0.0f;
            for (int i = 0; i < c->slots; ++i) sum += c->recent_ms[i];
            if (sum > c->outlier * (float)c->slots * ref) {
                ++c->st.outliers;
                c->recent_n = 0;
                if (c->incumbent != RRT_PATH_SINGLE) {              /* trial frames (the report may arrive after the trial's last frame) */
                    if (!c->trial_aborted && !c->decided) {
                        c->trial_aborted = true; ++c->st.trials_aborted; c->trial_every = std::min(8, c->trial_every * 2);
                        c->clean_from = frame + 1 + 2 * c->slots;   /* single-kernel frames already enqueued drain first */
                    }
                } else if (!c->demoted) {
                    c->demoted = true;
                    c->demoted_from = frame + 1;
                    c->trial_every = 2;
                }
                return RRT_OK;
            }
        }
    } else if (p != RRT_PATH_SINGLE) {
        c->recent_n = 0;
    }
    // ---- the samples
    if (trial_counted) {
        if ((int)c->trial_ms.size() < c->counted) c->trial_ms.push_back(ms);
    } else if (clean_incumbent) {
        if (c->trial_on && !c->decided && (int)c->block_b.size() < c->counted) c->block_b.push_back(ms);
        c->tail.push_back(ms);
        if ((int)c->tail.size() > c->counted) c->tail.erase(c->tail.begin());
    }
    if (p != RRT_PATH_SINGLE && (trial_counted || clean_incumbent)) {
        const std::vector<float>& v = trial_counted ? c->trial_ms : c->tail;
        if (v.size() >= (size_t)c->slots) c->st.last_three_pass_mean_ms = mean(v);
    }
    if (c->trial_on && !c->decided && !c->trial_aborted && (int)c->trial_ms.size() >= c->counted && (int)c->block_b.size() >= c->counted) decide(*c);
    return RRT_OK;
}

int rrt_path_chooser_get_stats(int id, rrt_path_chooser_stats* out) {
    std::lock_guard<std::mutex> lk(g_mu);
    Chooser* c = get(id);
    if (!c) return RRT_ERR_BAD_HANDLE;
    if (!out) return RRT_ERR_INVALID_ARGUMENT;
    *out = c->st;
    out->incumbent = c->incumbent;
    out->windows = c->win_index;
    return RRT_OK;
}

}  // extern "C"
