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
// Rule (per rank; the bytes do not depend on it):
//   - the sequence is cut into windows of `window_frames` frames; the INCUMBENT path renders a window, except for a TRIAL
//     of the other path at the window's start, 4 x frames_in_flight frames long, of which the first `frames_in_flight`
//     overlap the other path's frames and are not counted; the same number of incumbent frames after the trial is skipped
//     for the same reason;
//   - what is reported is a frame's interval on the rank: the time between the ends of consecutive frames' renders (with
//     frames in flight a frame's own start-to-end latency says nothing).  With n frames in flight on n streams these
//     intervals come in a pattern of period n (two frames end together, then a gap: measured, profiles/r06_chooser_probe.txt
//     -- the first version compared medians and took every other gap for an outlier), so what is COMPARED is the mean over a
//     multiple of n counted frames = elapsed time / frames, i.e. the sustained throughput of the path;
//   - the first window is short (4 x frames_in_flight + 1 frames) and has no trial: it measures the three-pass path;
//   - at the end of a window the path with the lower mean becomes the incumbent, if it wins by 4 % (hysteresis; the mean of nine
//     frames of a rank's share is good to about 2 %); a single kernel that does not win is tried again after 2, then 4 windows; with the
//     single kernel as incumbent every second window has its (three-pass) trial, so a single kernel that has become slow
//     is found within two windows;
//   - `frames_in_flight` consecutive single-kernel frames that together take more than `outlier` (1.5) x as long as the
//     three-pass path's mean ends a trial of the single kernel at once -- or, if the single kernel is the incumbent, hands
//     the rest of the window to the three-pass path -- and doubles the distance to the next trial (up to every 8th window).
// The reports arrive late (a frame's interval is known once it has been delivered, frames_in_flight frames after it was
// enqueued); the rule only needs them before the window ends.
#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/rrt.h"

namespace {

struct Chooser {
    int slots = 3, window = 48, trial = 6;
    float outlier = 1.5f, hysteresis = 0.96f;
    int incumbent = RRT_PATH_AUTO;
    int win_start = 1;               // first frame of the current window
    int win_index = 0;               // windows begun
    int trial_every = 1;             // incumbent three-pass: a trial of the single kernel at the start of every `trial_every`-th window
    int since_trial = 0;
    int first_window = 10;           // frames of the first window
    bool trial_on = false;           // this window has (had) a trial
    bool trial_aborted = false;
    int trial_end = 0;               // first frame after the trial
    bool demoted = false;            // the incumbent single kernel lost the rest of this window to the three-pass path
    int demoted_from = 0;
    std::vector<int8_t> policy_of;   // policy_of[frame % size]
    std::vector<float> ms[2];        // this window's counted intervals per policy (index: 0 = AUTO / three-pass, 1 = SINGLE)
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
    // the first window (a short one: create()) renders with the three-pass path only: the outlier rule needs its mean.
    // With the single kernel as incumbent every SECOND window has its (three-pass) trial -- 12 frames in 96 at the slower path,
    // and a single kernel that has become slow without tripping the outlier rule stays for two windows at most; trials OF
    // the single kernel back off after a lost or aborted one.
    c.trial_on = c.win_index > 1 && ++c.since_trial >= (c.incumbent == RRT_PATH_SINGLE ? 2 : c.trial_every);
    if (c.trial_on) c.since_trial = 0;
    c.trial_aborted = false;
    c.trial_end = frame + (c.trial_on ? c.trial : 0);
    c.demoted = false;
    c.recent_n = 0;
    c.ms[0].clear(); c.ms[1].clear();
    if (c.trial_on) ++c.st.trials;
}

void close_window(Chooser& c) {
    const int other = c.incumbent == RRT_PATH_SINGLE ? RRT_PATH_AUTO : RRT_PATH_SINGLE;
    const std::vector<float>& inc = c.ms[c.incumbent == RRT_PATH_SINGLE ? 1 : 0];
    const std::vector<float>& alt = c.ms[other == RRT_PATH_SINGLE ? 1 : 0];
    if (c.demoted) {                                  // the single kernel met a wavefront that outlasts the frames in flight
        c.incumbent = RRT_PATH_AUTO;
        ++c.st.switches;
        return;
    }
    if (!c.trial_on || c.trial_aborted || inc.size() < (size_t)c.slots || alt.size() < (size_t)c.slots) return;
    const float mi = mean(inc), ma = mean(alt);
    if (ma < c.hysteresis * mi) {
        c.incumbent = other;
        ++c.st.switches;
        c.trial_every = 1;
    } else if (c.incumbent == RRT_PATH_AUTO) {
        c.trial_every = std::min(4, c.trial_every * 2);       // the single kernel did not win: look again in two windows, then in four
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
    c->window = window_frames > 0 ? window_frames : 48;
    if (c->window < 2 * c->trial + 2 * c->slots) c->window = 2 * c->trial + 2 * c->slots;      // room for counted frames of both paths
    c->first_window = 4 * frames_in_flight + 1;
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
    else if (frame >= c->win_start + (c->win_index == 1 ? c->first_window : c->window)) { close_window(*c); begin_window(*c, frame); }
    const int other = c->incumbent == RRT_PATH_SINGLE ? RRT_PATH_AUTO : RRT_PATH_SINGLE;
    int p = c->incumbent;
    if (c->trial_on && !c->trial_aborted && frame < c->trial_end) p = other;
    if (c->demoted && frame >= c->demoted_from) p = RRT_PATH_AUTO;
    c->policy_of[(size_t)frame % c->policy_of.size()] = (int8_t)p;
    ++c->st.frames[p == RRT_PATH_SINGLE ? 1 : 0];
    *policy_out = p;
    return RRT_OK;
}

/* the sustained time of frame `frame` on this rank: milliseconds between the end of frame `frame - 1`'s render and its own */
int rrt_path_chooser_report(int id, int frame, float ms) {
    std::lock_guard<std::mutex> lk(g_mu);
    Chooser* c = get(id);
    if (!c) return RRT_ERR_BAD_HANDLE;
    if (frame < 1 || !(ms >= 0.0f)) return RRT_ERR_INVALID_ARGUMENT;
    if (c->win_index == 0 || frame < c->win_start) return RRT_OK;              // a report of an earlier window: too late to matter
    const int p = c->policy_of[(size_t)frame % c->policy_of.size()];
    // frames whose neighbours in flight ran the other path do not count: the first `slots` of a window (its trial, or the frames
    // after the previous window's) and the first `slots` after the trial
    const int rel = frame - c->win_start;
    const bool in_trial = c->trial_on && frame < c->trial_end;
    const bool mixed = rel < c->slots || (c->trial_on && !in_trial && frame < c->trial_end + c->slots);
    if (p == RRT_PATH_SINGLE && !mixed) {
        // the outlier rule on `slots` consecutive single-kernel frames together (one period of the interval pattern)
        if (c->recent_n > 0 && frame != c->recent_last + 1) c->recent_n = 0;
        if (c->recent_n == c->slots) { for (int i = 1; i < c->slots; ++i) c->recent_ms[i - 1] = c->recent_ms[i]; --c->recent_n; }
        c->recent_ms[c->recent_n++] = ms;
        c->recent_last = frame;
        const float ref = c->ms[0].size() >= (size_t)c->slots ? mean(c->ms[0]) : c->st.last_three_pass_mean_ms;
        if (c->recent_n == c->slots && ref > 0.0f) {
            float sum = 0.0f;
            for (int i = 0; i < c->slots; ++i) sum += c->recent_ms[i];
            if (sum > c->outlier * (float)c->slots * ref) {
                ++c->st.outliers;
                c->recent_n = 0;
                if (c->incumbent != RRT_PATH_SINGLE) {              /* trial frames (the report may arrive after the trial's last frame) */
                    if (!c->trial_aborted) { c->trial_aborted = true; ++c->st.trials_aborted; c->trial_every = std::min(8, c->trial_every * 2); }
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
    if (!mixed) {
        c->ms[p == RRT_PATH_SINGLE ? 1 : 0].push_back(ms);
        if (p != RRT_PATH_SINGLE && c->ms[0].size() >= (size_t)c->slots) c->st.last_three_pass_mean_ms = mean(c->ms[0]);
    }
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
