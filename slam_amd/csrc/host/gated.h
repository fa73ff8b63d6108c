// Gated per-particle association reduced to ONE association per step (slam-backend -assoc gated).
//
// Every particle gates every observation against its own map with EKF-SLAM's rule (EKFSLAM::dataAssociate,
// algorithms/ekfslam.cpp:151-189: nearest neighbour inside GATE_REJECT, a new feature beyond GATE_AUGMENT, nothing in between:
// slamgpu_associate, decision by decision the reference's); the device layout keeps one landmark count for all particles (the
// reference's FastSLAM wrappers associate per step too: fastslam2wrapper.cpp:84), so the particles' labels become one label per
// observation by weighted vote.  What round 5 did with the vote -- take the label with the largest share, open a landmark
// whenever that label was "new" -- was right decision by decision and wrong run by run: a landmark opened by a 40 % plurality
// exists for EVERY particle from then on, competes with the real one for its observations, drags the poses, and more "new"
// votes follow (22 of 40 whole runs ended at the map's capacity: profiles/gated_association_whole_runs_r05.txt).
//
// The policy here (round 6) is the reference's own caution applied to the vote, plus a map that can shrink:
//   * a landmark is OPENED only on a supermajority of the particle weight saying "new" (new_share): between GATE_REJECT and
//     GATE_AUGMENT the reference discards an observation rather than guess (ekfslam.cpp:182-186); a split vote is the same doubt;
//   * an observation is MATCHED only when its label carries match_share of the weight;
//   * every landmark holds a CREDIT: +1 when it is matched, -1 on a step on which it should have been seen from the pose of the
//     particle the map is read from -- inside the sensor's range and field of view with a margin (getObservations,
//     core.cpp:185-273: the half disc in front of the vehicle) -- and was not; a landmark whose credit falls below retire_below is
//     RETIRED (slamgpu_retire_landmarks: no particle gates against it any more).  A real landmark in view is matched step after
//     step and sits at the cap; of a real landmark and its duplicate only one is matched per step, the other pays.
// The reference's Particle could drop an entry (landmarkXs / landmarkPs are per-particle vectors: Particle.cpp:61-73); a shared
// landmark count cannot renumber in the middle of a run, so a retired landmark stays in the maps, inert.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <cstdint>
#include <vector>

namespace slamhost {

struct GatedPolicy {
    // tunables (slam-backend: -ASSOC_NEW_SHARE, -ASSOC_MATCH_SHARE, -ASSOC_CREDIT_START, -ASSOC_CREDIT_MAX, -ASSOC_RETIRE_BELOW;
    // -ASSOC_POLICY 0 = round 5's plurality vote without credits)
    bool enabled = true;
    float new_share = 0.9f, match_share = 0.5f;
    int credit_start = 1, credit_max = 6, retire_below = -3;
    float range_margin = 0.85f, front_margin = 1.0f;  // "should have been seen": d < range_margin * MAX_RANGE, > front_margin m ahead

    // second stage (see step()): an observation the vote leaves without a landmark is looked at in WORLD coordinates, from the pose
    // and map of the particle the credits are kept against: rescue_base + rescue_per_m * range is the distance within which a mapped
    // landmark still explains it; the match is taken only if it is unique (the second-nearest landmark at least unique_ratio times
    // as far and beyond the radius, where "landmark" includes the points this step's OTHER observations imply); a landmark is not
    // opened when exactly ONE candidate lies within new_factor radii (where several do, the surroundings are dense against the
    // radius -- BASELINE config 5: 10 000 landmarks, ~2 m apart -- and the gates decide alone)
    float rescue_base = 2.0f, rescue_per_m = 0.05f, unique_ratio = 2.0f, new_factor = 2.0f;
    bool rescue = true;
    bool grid = true;  // the second stage's candidates from a uniform grid (false: every landmark and point, in one cell: the same decisions, tests)
    int n_rescued = 0, n_new_refused = 0;

    std::vector<int32_t> decision;  // of the last step, per observation: landmark index, -1 opened as new, -2 left unused; +1000000 = by the second stage
    std::vector<int> credit;    // per landmark, retired ones included
    std::vector<char> retired;
    int n_opened = 0, n_retired = 0, n_discarded_votes = 0;

    int active() const { return (int) credit.size() - n_retired; }

    // One step.  z[2 nz], the vote (consensus / support of slamgpu_associate), the map of ONE particle (its pose xv[3] and landmark
    // means xf[2 nf], nf = the context's landmark count before this step), the sensor's range, room = landmarks that may still be
    // opened.  Out: the update's packet (zf / idf / zn) and the landmarks to retire now.
    void step(const float *z, int nz, const int32_t *consensus, const float *support, const float *xv, const float *xf, int nf, float max_range,
              int room, std::vector<float> &zf, std::vector<int32_t> &idf, std::vector<float> &zn, std::vector<int32_t> &retire) {
        zf.clear();
        idf.clear();
        zn.clear();
        retire.clear();
        credit.resize((size_t) nf, credit_start);
        retired.resize((size_t) nf, 0);
        std::vector<char> matched((size_t) nf, 0);
        std::vector<int> pending;  // observations the vote gave no landmark
        decision.assign((size_t) nz, -2);
        for (int q = 0; q < nz; q++) {
            const int32_t lab = consensus[q];
            const float sh = support ? support[q] : 1.0f;
            if (lab >= 0 && lab < nf && !retired[(size_t) lab] && (!enabled || sh >= match_share)) {
                zf.push_back(z[2 * q]);
                zf.push_back(z[2 * q + 1]);
                idf.push_back(lab);
                matched[(size_t) lab] = 1;
                decision[(size_t) q] = lab;
            } else if (!enabled) {
                if (lab == -1 /* SLAMGPU_ASSOC_NEW */ && (int) (zn.size() / 2) < room) {
                    zn.push_back(z[2 * q]);
                    zn.push_back(z[2 * q + 1]);
                    decision[(size_t) q] = -1;
                }
            } else {
                pending.push_back(q);
            }
        }
        // Second stage.  The per-particle gates are chi-square tests against the particle's OWN landmark covariance: they know nothing
        // of a drift every particle shares (a depleted set after a long stretch without re-observation: closing the loop of
        // example_webmap, landmarks come back into view at 60 m with the whole set ~1 m off along the line of sight: ten standard
        // deviations of the range sensor, a unanimous "new" -- and a duplicate of every landmark of the first lap), and between
        // GATE_REJECT and GATE_AUGMENT they drop exactly the observations that contradict the estimate, so an error feeds itself.
        // The reference's EKF carries the pose covariance in its innovation (ekfslam.cpp:160-170) and widens the gates with the
        // drift; a particle set that has lost its spread cannot.  What stands in for it here: where the map is sparse against the
        // drift the geometry alone decides -- the mapped landmark nearest to the point the observation implies, if it is the only one
        // anywhere near.
        // (the points ALL of this step's observations imply: where the environment is dense against the radius -- BASELINE config 5:
        // 10 000 landmarks ~2 m apart, 860 observations a step -- two observations land near one mapped landmark, the geometry cannot
        // tell a drifted re-observation from a neighbour nobody has mapped yet, and the gates' verdict stands: an observation's
        // nearest OTHER observation counts as a second candidate in the tests below)
        std::vector<float> ox((size_t) nz), oy((size_t) nz);
        for (int q = 0; q < nz; q++) {
            ox[(size_t) q] = xv[0] + z[2 * q] * std::cos(xv[2] + z[2 * q + 1]);
            oy[(size_t) q] = xv[1] + z[2 * q] * std::sin(xv[2] + z[2 * q + 1]);
        }
        // (round 6, end: the candidates of a pending observation come from a uniform grid over the mapped landmarks and over this step's
        // points instead of a scan of all of them -- 250 pending observations x (9 814 landmarks + 865 points) with a square root each
        // were 2.7 ms of a 5.4 ms step on the 10 000-landmark map.  Every test below compares d1 / d2 with bounds of at most
        // B = max(unique_ratio, new_factor, 1) x rho; the grid's cells are at least the largest B wide, so the 3 x 3 cells around the
        // point hold everything within B, candidates are visited in index order (ties as before), and a distance beyond B reads as
        // "further than B" either way: the decisions are those of the full scan.)
        struct Grid {
            float x0 = 0, y0 = 0, inv = 0;
            int nx = 1, ny = 1;
            std::vector<int32_t> start, item;
            void build(const float *px, const float *py, int stride, int n, const std::vector<char> *skip, float cs, float bx0, float by0, float bx1, float by1) {
                x0 = bx0;
                y0 = by0;
                inv = 1.0f / cs;
                nx = std::max(1, std::min(1024, (int) ((bx1 - bx0) * inv) + 1));
                ny = std::max(1, std::min(1024, (int) ((by1 - by0) * inv) + 1));
                start.assign((size_t) nx * ny + 1, 0);
                std::vector<int32_t> cell((size_t) n, -1);
                for (int k = 0; k < n; k++) {
                    if (skip && (*skip)[(size_t) k]) continue;
                    const float x = px[(size_t) k * stride], y = py[(size_t) k * stride];
                    if (!(x == x) || !(y == y)) continue;
                    cell[(size_t) k] = cy(y) * nx + cx(x);
                    start[(size_t) cell[(size_t) k] + 1]++;
                }
                for (size_t c = 0; c + 1 < start.size(); c++) start[c + 1] += start[c];
                item.assign((size_t) start.back(), 0);
                std::vector<int32_t> fill(start.begin(), start.end() - 1);
                for (int k = 0; k < n; k++)  // (ascending k inside every cell)
                    if (cell[(size_t) k] >= 0) item[(size_t) fill[(size_t) cell[(size_t) k]]++] = k;
            }
            int cx(float x) const { return std::max(0, std::min(nx - 1, (int) std::floor((x - x0) * inv))); }
            int cy(float y) const { return std::max(0, std::min(ny - 1, (int) std::floor((y - y0) * inv))); }
            // the members of the 3 x 3 cells around (x, y), ascending
            void around(float x, float y, std::vector<int32_t> &out) const {
                out.clear();
                const int ax = cx(x), ay = cy(y);
                for (int yy = std::max(0, ay - 1); yy <= std::min(ny - 1, ay + 1); yy++)
                    for (int xx = std::max(0, ax - 1); xx <= std::min(nx - 1, ax + 1); xx++)
                        out.insert(out.end(), item.begin() + start[(size_t) yy * nx + xx], item.begin() + start[(size_t) yy * nx + xx + 1]);
                std::sort(out.begin(), out.end());
            }
        };
        Grid glm, gob;
        std::vector<int32_t> cand;
        if (!pending.empty()) {
            float rmax = 0.0f, bx0 = INFINITY, by0 = INFINITY, bx1 = -INFINITY, by1 = -INFINITY;
            for (int q : pending) rmax = std::max(rmax, std::fabs(z[2 * q]));
            for (int q = 0; q < nz; q++) {
                bx0 = std::min(bx0, ox[(size_t) q]); bx1 = std::max(bx1, ox[(size_t) q]);
                by0 = std::min(by0, oy[(size_t) q]); by1 = std::max(by1, oy[(size_t) q]);
            }
            for (int j = 0; j < nf; j++) {
                if (retired[(size_t) j] || !(xf[2 * j] == xf[2 * j])) continue;
                bx0 = std::min(bx0, xf[2 * j]); bx1 = std::max(bx1, xf[2 * j]);
                by0 = std::min(by0, xf[2 * j + 1]); by1 = std::max(by1, xf[2 * j + 1]);
            }
            // (cells at least the largest B wide -- and wide enough that the grid has at most 1 024 x 1 024 of them; a cell wider than
            // B only means more candidates)
            float cs = std::max(1e-3f, std::max(std::max(unique_ratio, new_factor), 1.0f) * (rescue_base + rescue_per_m * rmax)) * 1.0001f;
            cs = std::max(cs, std::max(bx1 - bx0, by1 - by0) / 1000.0f);
            if (!grid) cs = 4.0f * (std::max(bx1 - bx0, by1 - by0) + 1.0f);  // (one cell)
            glm.build(xf, xf + 1, 2, nf, &retired, cs, bx0, by0, bx1, by1);
            gob.build(ox.data(), oy.data(), 1, nz, nullptr, cs, bx0, by0, bx1, by1);
        }
        for (int q : pending) {
            const float r = z[2 * q], b = z[2 * q + 1];
            const float px = ox[(size_t) q], py = oy[(size_t) q];
            float d1 = INFINITY, d2 = INFINITY;
            int j1 = -1;
            gob.around(px, py, cand);
            for (int o : cand) {
                if (o == q) continue;
                const float dx = ox[(size_t) o] - px, dy = oy[(size_t) o] - py;
                d2 = std::min(d2, std::sqrt(dx * dx + dy * dy));
            }
            glm.around(px, py, cand);
            for (int j : cand) {
                const float dx = xf[2 * j] - px, dy = xf[2 * j + 1] - py;
                const float d = std::sqrt(dx * dx + dy * dy);
                if (d < d1) {
                    d2 = std::min(d2, d1);
                    d1 = d;
                    j1 = j;
                } else if (d < d2) {
                    d2 = d;
                }
            }
            const float rho = rescue_base + rescue_per_m * std::fabs(r);
            const float sh = support ? support[q] : 1.0f;
            if (rescue && j1 >= 0 && d1 <= rho && d2 > rho && d2 >= unique_ratio * d1 && !matched[(size_t) j1]) {
                zf.push_back(r);
                zf.push_back(b);
                idf.push_back(j1);
                matched[(size_t) j1] = 1;
                decision[(size_t) q] = 1000000 + j1;
                n_rescued++;
            } else if (consensus[q] == -1 /* SLAMGPU_ASSOC_NEW */ && sh >= new_share && (int) (zn.size() / 2) < room) {
                // not next to a mapped landmark -- unless the surroundings are dense (a second mapped landmark, or another of this
                // step's observations, that close): then the gates' verdict stands
                if (!rescue || d1 > new_factor * rho || d2 <= new_factor * rho) {
                    zn.push_back(r);
                    zn.push_back(b);
                    decision[(size_t) q] = -1;
                } else {
                    n_new_refused++;
                }
            } else {
                n_discarded_votes++;
            }
        }
        n_opened += (int) (zn.size() / 2);
        if (!enabled) return;
        for (int j = 0; j < nf; j++) {
            if (retired[(size_t) j]) continue;
            if (matched[(size_t) j]) {
                credit[(size_t) j] = std::min(credit_max, credit[(size_t) j] + 1);
                continue;
            }
            const float dx = xf[2 * j] - xv[0], dy = xf[2 * j + 1] - xv[1];
            const float d = std::sqrt(dx * dx + dy * dy), ahead = dx * std::cos(xv[2]) + dy * std::sin(xv[2]);
            if (d < range_margin * max_range && ahead > front_margin) {
                if (--credit[(size_t) j] < retire_below) {
                    retired[(size_t) j] = 1;
                    n_retired++;
                    retire.push_back(j);
                }
            }
        }
    }
};

}  // namespace slamhost
