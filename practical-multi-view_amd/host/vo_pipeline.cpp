// See vo_pipeline.h. Line references are into /root/reference/.
#include "vo_pipeline.h"
#include "vo_math.h"
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <atomic>
#include <exception>
#include <cfloat>

namespace vo {

// ---- Feature::Hasher support: hash<string>(to_string(v)) memoised ------------------------------------------
size_t coord_hash(int v) {
    static const int LO = -4096, HI = 16384;
    static std::vector<size_t>* table = [] {
        auto* t = new std::vector<size_t>(HI - LO);
        for (int i = LO; i < HI; i++) (*t)[i - LO] = std::hash<std::string>{}(std::to_string(i));
        return t;
    }();
    if (v >= LO && v < HI) return (*table)[v - LO];
    return std::hash<std::string>{}(std::to_string(v));
}

void rodrigues_v2m(const double r[3], double R[9]) { vmath::rodrigues_v2m(r, R); }
void rodrigues_m2v(const double R[9], double r[3]) { vmath::rodrigues_m2v(R, r); }

// ---- OpenCVGoodFeatureExtractor.cpp:4-21 -------------------------------------------------------------------
static std::vector<Feature> corners_to_features(const std::vector<std::pair<int, int>>& c, const std::vector<double>* score,
                                                Feature::extractor det) {
    std::vector<Feature> feats;
    for (size_t i = 0; i < c.size(); i++) {
        Feature f;                 // default ctor: tracked = false
        f.row = c[i].second;
        f.column = c[i].first;
        f.detector = det;
        if (det == Feature::cv_good) f.tracked = true;   // :16 (ShiTomasi leaves tracked=false, its score is set instead)
        if (score) f.score = (*score)[i];
        feats.push_back(f);
    }
    return feats;
}
std::vector<Feature> GoodFeatureExtractorBase::extractFeatures(Frame& src, int max) {
    std::vector<ImageView> cells{src.bw};
    std::vector<std::vector<std::pair<int, int>>> out;
    gftt(cells, max, out);
    return corners_to_features(out[0], nullptr, Feature::cv_good);
}
std::vector<std::vector<Feature>> GoodFeatureExtractorBase::extractGrid(std::vector<Frame>& cells, int max) {
    std::vector<ImageView> views;
    for (auto& c : cells) views.push_back(c.bw);
    std::vector<std::vector<std::pair<int, int>>> out;
    gftt(views, max, out);
    std::vector<std::vector<Feature>> res;
    for (auto& o : out) res.push_back(corners_to_features(o, nullptr, Feature::cv_good));
    return res;
}
// ---- ShiTomasiFeatureExtractor.cpp:5-47 --------------------------------------------------------------------
std::vector<Feature> ShiTomasiExtractorBase::extractFeatures(Frame& src, int max) {
    std::vector<ImageView> cells{src.bw};
    std::vector<std::vector<std::pair<int, int>>> out;
    std::vector<std::vector<double>> sc;
    shitomasi(cells, max, out, sc);
    return corners_to_features(out[0], &sc[0], Feature::shi_tomasi);
}
std::vector<std::vector<Feature>> ShiTomasiExtractorBase::extractGrid(std::vector<Frame>& cells, int max) {
    std::vector<ImageView> views;
    for (auto& c : cells) views.push_back(c.bw);
    std::vector<std::vector<std::pair<int, int>>> out;
    std::vector<std::vector<double>> sc;
    shitomasi(views, max, out, sc);
    std::vector<std::vector<Feature>> res;
    for (size_t i = 0; i < out.size(); i++) res.push_back(corners_to_features(out[i], &sc[i], Feature::shi_tomasi));
    return res;
}

// ---- OpenCVFASTFeatureExtractor.cpp:4-21 ---------------------------------------------------------------------------
static std::vector<Feature> keypoints_to_features(const std::vector<std::pair<int, int>>& kp, const std::vector<float>& resp) {
    std::vector<Feature> feats;
    for (size_t i = 0; i < kp.size(); i++) {
        Feature f(kp[i].first, kp[i].second);   // Feature(k.pt): column = x, row = y; tracked = true (:16), detector stays cv_good
        f.score = resp[i];                      // k.response
        f.tracked = true;
        feats.push_back(f);
    }
    return feats;
}
std::vector<Feature> FastExtractorBase::extractFeatures(Frame& src, int max) {
    std::vector<ImageView> cells{src.bw};
    std::vector<std::vector<std::pair<int, int>>> out;
    std::vector<std::vector<float>> resp;
    fast(cells, max, out, resp);
    return keypoints_to_features(out[0], resp[0]);
}
std::vector<std::vector<Feature>> FastExtractorBase::extractGrid(std::vector<Frame>& cells, int max) {
    std::vector<ImageView> views;
    for (auto& c : cells) views.push_back(c.bw);
    std::vector<std::vector<std::pair<int, int>>> out;
    std::vector<std::vector<float>> resp;
    fast(views, max, out, resp);
    std::vector<std::vector<Feature>> res;
    for (size_t i = 0; i < out.size(); i++) res.push_back(keypoints_to_features(out[i], resp[i]));
    return res;
}

// ---- kNNFeatureMatcher.cpp:3-61 --------------------------------------------------------------------------------------
fmap KnnFeatureMatcherBase::matchFeatures(Frame& src, Frame& next) {
    fmap map;
    double avg = 0;
    std::vector<std::shared_ptr<Feature>> new_feats, old_feats;
    std::vector<bool> tracked;
    std::vector<Feature> cmp_feats = extractor->extractFeatures(next, 1000);   // :11, on the whole next frame
    for (auto& p : src.map) old_feats.push_back(p.first);
    const int n = (int)old_feats.size(), m = (int)cmp_feats.size();
    std::vector<int> src_xy(2 * (size_t)n), cmp_xy(2 * (size_t)m), best(n, -1);
    std::vector<float> errs(n, 0.f);
    for (int i = 0; i < n; i++) { src_xy[2 * i] = old_feats[i]->column; src_xy[2 * i + 1] = old_feats[i]->row; }
    for (int j = 0; j < m; j++) { cmp_xy[2 * j] = cmp_feats[j].column; cmp_xy[2 * j + 1] = cmp_feats[j].row; }
    if (n > 0) knn(src.bw, next.bw, src_xy.data(), n, cmp_xy.data(), m, best.data(), errs.data());   // :15-31 for every feature
    for (int i = 0; i < n; i++) {
        Feature best_fit = best[i] >= 0 ? cmp_feats[best[i]] : Feature();   // the neighbour copy (default-constructed when there was none)
        const float err = errs[i];
        if (err < threshold) {
            tracked.push_back(true);
            best_fit.tracked = true;
            best_fit.displacement = old_feats[i]->distance(best_fit);
            avg += best_fit.displacement;
        } else
            tracked.push_back(false);   // (the copy keeps the candidate's own `tracked` flag: quirk of the reference, :32-41)
        new_feats.push_back(std::make_shared<Feature>(best_fit));
    }
    avg /= (double)new_feats.size();
    for (auto& f : new_feats)
        if (f->displacement > 3 * avg) f->tracked = false;
    for (size_t i = 0; i < old_feats.size(); i++)
        if (new_feats[i]->tracked) {
            next.map[new_feats[i]] = src.map[old_feats[i]];
            map[old_feats[i]] = new_feats[i];
        }
    return map;
}

// ---- OpenCVLucasKanadeFM.cpp:5-32 ----------------------------------------------------------------------------
fmap LucasKanadeFMBase::matchFeatures(Frame& src, Frame& next) {
    fmap correspondences;
    std::vector<float> prev_points, next_points;
    for (auto const& p : src.map) {
        prev_points.push_back((float)p.first->column);
        prev_points.push_back((float)p.first->row);
    }
    const int n = (int)(prev_points.size() / 2);
    next_points.resize(prev_points.size());
    std::vector<uint8_t> status(n);
    std::vector<float> err(n);
    if (n > 0) pyrlk(src.bw, next.bw, prev_points.data(), n, next_points.data(), status.data(), err.data());
    int i = 0;
    for (auto const& p : src.map) {
        if ((int)status.size() < i + 1) continue;
        if (status[i]) {
            // Feature(next_points[i].x, next_points[i].y): float -> int truncation toward zero
            std::shared_ptr<Feature> f = std::make_shared<Feature>(Feature((int)next_points[2 * i], (int)next_points[2 * i + 1]));
            next.map[f] = p.second;
            correspondences[p.first] = f;
        }
        i++;
    }
    return correspondences;
}

// ---- OpenCVEPnPSolver.cpp:4-50 ---------------------------------------------------------------------------------
// front-end: remember, in the key object of every feat_corr entry, where the entry lives (see Feature::corr_slot)
static void index_feat_corr(Frame& fr) {
    for (auto& p : fr.feat_corr)
        if (std::shared_ptr<Feature> k = p.first.lock()) { k->corr_slot = &p.second; k->corr_owner = &fr.feat_corr; k->corr_feat = p.second.lock().get(); }
}

// front-end: the gather list of solvePnP(src, next) (see PnPLink). `next` must be at its final address with all its features in.
static void build_pnp_links(Frame& src, Frame& next) {
    for (auto& p : next.map) { p.first->map_slot = &p.second; p.first->map_owner = &next.map; }
    src.pnp_links.clear();
    src.pnp_links.reserve(src.map.size());
    for (auto& p : src.map) {
        PnPLink L;
        L.src_val = &p.second; L.key = &p.first;
        L.key_column = p.first->column; L.key_row = p.first->row;
        if (p.first->corr_owner == (const void*)&src.feat_corr) {   // this very object is the key of its feat_corr entry
            Feature* f = p.first->corr_feat;
            if (!f) L.kind = 0;
            else if (f->map_owner == (const void*)&next.map) { L.kind = 1; L.next_slot = f->map_slot; L.f = f; }
            else {   // not a key object of next.map: an equal key's node is where next.map[f] lands; no such node -> operator[] would insert
                std::shared_ptr<Feature> fs = p.first->corr_slot->lock();
                auto it = fs ? next.map.find(fs) : next.map.end();
                if (it != next.map.end()) { L.kind = 1; L.next_slot = &it->second; L.f = f; }
            }
        }
        src.pnp_links.push_back(L);
    }
    src.pnp_links_for = &next.map;
    // the two-view gather (OpenCVFivePointTri.cpp:9-22) and where its landmarks go (:47-50)
    for (auto& p : src.map) { p.first->map_slot = &p.second; p.first->map_owner = &src.map; }   // (src's hints pointed into the map of its own build step)
    src.tri_links.clear();
    src.tri_links.reserve(src.feat_corr.size());
    bool complete = true;
    for (auto& p : src.feat_corr) {
        if (p.first.expired() || p.second.expired()) continue;
        std::shared_ptr<Feature> fst = p.first.lock(), sec = p.second.lock();
        TriLink T;
        T.fst = fst.get(); T.sec = sec.get();
        if (fst->map_owner == (const void*)&src.map) T.src_slot = fst->map_slot;
        else { auto it = src.map.find(fst); if (it != src.map.end()) T.src_slot = &it->second; }
        if (sec->map_owner == (const void*)&next.map) T.next_slot = sec->map_slot;
        else { auto it = next.map.find(sec); if (it != next.map.end()) T.next_slot = &it->second; }
        if (!T.src_slot || !T.next_slot) complete = false;   // operator[] would insert a node: leave this pair to the reference's loop
        src.tri_links.push_back(T);
    }
    src.tri_links_src = complete ? (const void*)&src.map : nullptr;
}

void EPnPSolverBase::solvePnP(Frame& src, Frame& next, Mat3& R_out, Vec3& t_out) {
    const int j = src.frame;
    std::vector<float> obj_points, img_points;
    double _R_rod[3];
    rodrigues_m2v(R_out.m, _R_rod);
    std::vector<std::weak_ptr<Feature3D>> local_feats3d;
    HostProfScope* hps = new HostProfScope(tracker->stats.hp.t[0]);
    obj_points.reserve(3 * src.map.size()); img_points.reserve(2 * src.map.size()); local_feats3d.reserve(src.map.size());
    const auto take = [&](std::shared_ptr<Feature3D>& f3d, const Feature* f) {   // :22-27 for one landmark / image point pair
        f3d->transformInv(tracker->R[j], tracker->t[j]);
        float px = f3d->x, py = f3d->y, pz = f3d->z;
        pz *= -1;
        obj_points.push_back(px); obj_points.push_back(py); obj_points.push_back(pz);
        img_points.push_back((float)f->column); img_points.push_back((float)f->row);
        f3d->transform(tracker->R[j], tracker->t[j]);       // float round trip (quirk Q7)
        local_feats3d.push_back(std::move(f3d));
    };
    // one src.map entry the way the reference walks it. src.feat_corr[p.first]: the entry found by coordinate equality. If this very
    // object is the entry's key the front-end left its address in corr_slot; otherwise (same-pixel twin, or no correspondence) look it
    // up as the reference does - operator[] then inserts the empty entry of quirk Q10.
    const auto slow_entry = [&](const std::shared_ptr<Feature>& key, std::shared_ptr<Feature3D>& f3d) {
        Feature* f;
        if (key->corr_owner == (const void*)&src.feat_corr) {
            f = key->corr_feat;                       // no reference-count traffic on the (front-end-created) feature
            if (!f) return;
            if (f->map_owner == (const void*)&next.map) *f->map_slot = std::weak_ptr<Feature3D>(f3d);
            else next.map[key->corr_slot->lock()] = std::weak_ptr<Feature3D>(f3d);
        } else {
            std::shared_ptr<Feature> fs = src.feat_corr[key].lock();
            if (!fs) return;
            f = fs.get();
            if (f->map_owner == (const void*)&next.map) *f->map_slot = std::weak_ptr<Feature3D>(f3d);
            else next.map[fs] = std::weak_ptr<Feature3D>(f3d);
        }
        take(f3d, f);
    };
    if (src.pnp_links_for == (const void*)&next.map && src.pnp_links.size() == src.map.size()) {
        // the front-end's list: same entries in the same order, only the landmark's liveness is looked up here
        for (const PnPLink& L : src.pnp_links) {
            std::shared_ptr<Feature3D> f3d = L.src_val->lock();    // (expired() + lock() in the reference: one atomic round trip here)
            if (!f3d) continue;
            if (L.kind == 1) { *L.next_slot = std::weak_ptr<Feature3D>(f3d); take(f3d, L.f); }
            else if (L.kind == 2) slow_entry(*L.key, f3d);
        }
    } else {
        for (auto& p : next.map) { p.first->map_slot = &p.second; p.first->map_owner = &next.map; }   // next.map[f] below without hashing
        for (auto& p : src.map) {
            std::shared_ptr<Feature3D> f3d = p.second.lock();
            if (!f3d) continue;
            slow_entry(p.first, f3d);
        }
    }
    delete hps;
    std::vector<int> inliers;
    const int m = (int)(obj_points.size() / 3);
    tracker->stats.pnp_calls++; tracker->stats.pnp_points += m;
    {
        const auto k0 = std::chrono::steady_clock::now();
        pnp_ransac(obj_points.data(), img_points.data(), m, tracker->camera, _R_rod, t_out.v, inliers);
        tracker->stats.t_pnp_kernel += std::chrono::duration<double>(std::chrono::steady_clock::now() - k0).count();
    }
    rodrigues_v2m(_R_rod, R_out.m);
    HostProfScope hps2(tracker->stats.hp.t[1]);
    // Removing RANSAC outliers (:40-49)
    std::vector<uint8_t> is_inlier(m, 0);
    for (int i : inliers) if (i >= 0 && i < m) is_inlier[i] = 1;
    for (int i = 0; i < m; i++) {
        if (!is_inlier[i]) {
            if (local_feats3d[i].expired()) continue;
            std::shared_ptr<Feature3D> f3d = local_feats3d[i].lock();
            tracker->feats3d.erase(f3d->self);
        }
    }
}

// ---- CeresBundleAdjustment.cpp:5-89 ------------------------------------------------------------------------------
void BundleAdjustmentBase::apply(Frame& f) {
    const int fn = (int)f.frame + 1;
    const int n = std::min(tracker->cfg.bundle_size, fn);
    // Work vectors: local, sized from the previous solve so that they do not regrow element by element. (Keeping the buffers themselves
    // between calls was measured and dropped: 6 ms per step for one sequence, but -7 % in the batched leg - 128 sequences x 50 KB of
    // buffers that are touched every other frame instead of memory the allocator hands straight back to the next gather.)
    std::vector<int> cam_frame, obs_cam, obs_pt;      // window frames in order (skipping 0); camera / point index per residual block
    std::vector<double> tr_opt, obs;                  // 6 per window frame; 2 per residual block
    cam_frame.reserve((size_t)n); tr_opt.reserve((size_t)6 * n);
    obs.reserve(2 * last_obs + 64); obs_cam.reserve(last_obs + 32); obs_pt.reserve(last_obs + 32);
    const unsigned epoch = ++epoch_counter;   // p3d_index of the reference, as epoch-stamped arrays over the landmark ids
    std::vector<Feature3D*> p3d_ptr;                  // (raw: nothing erases a landmark between the gather and the update below; one thread)
    std::vector<double> p3d_opt;
    p3d_ptr.reserve(last_points + 32); p3d_opt.reserve(3 * last_points + 96);
    HostProfScope* hpg = new HostProfScope(tracker->stats.hp.t[2]);
    std::vector<std::shared_ptr<Frame>> window((size_t)n);   // snapshot under the lock: the front-end thread may be appending
    {
        std::lock_guard<std::mutex> lk(tracker->frames_mu);
        for (int i = fn - n; i < fn; i++) window[(size_t)(i - (fn - n))] = tracker->frames[i];
    }
    for (int i = fn - n; i < fn; i++) {
        if (i == 0) continue;
        const std::shared_ptr<Frame>& frame = window[(size_t)(i - (fn - n))];
        double rod[3];
        Mat3 Rt = tracker->R[i].t();
        rodrigues_m2v(Rt.m, rod);
        const int ci = (int)cam_frame.size();
        cam_frame.push_back(i);
        tr_opt.push_back(rod[0]); tr_opt.push_back(rod[1]); tr_opt.push_back(rod[2]);
        tr_opt.push_back(-tracker->t[i].v[0]); tr_opt.push_back(-tracker->t[i].v[1]); tr_opt.push_back(-tracker->t[i].v[2]);
        const bool flat = frame->links_cover_map();   // the front-end's list holds the same entries in the same order
        const size_t n_entries = flat ? frame->pnp_links.size() : 0;
        auto entry = frame->map.begin();
        for (size_t e = 0; flat ? e < n_entries : entry != frame->map.end(); flat ? (void)++e : (void)++entry) {
            std::shared_ptr<Feature3D> f3d = flat ? frame->pnp_links[e].src_val->lock() : entry->second.lock();
            if (!f3d) continue;
            const int ft_column = flat ? frame->pnp_links[e].key_column : entry->first->column;
            const int ft_row = flat ? frame->pnp_links[e].key_row : entry->first->row;
            obs.push_back((double)ft_column); obs.push_back((double)ft_row);
            // index of the landmark in first-seen order (p3d_opt of the reference); landmark ids are dense creation numbers
            const size_t lid = (size_t)f3d->id;
            if (lid >= seen_epoch.size()) { seen_epoch.resize(lid + 4096, 0); seen_index.resize(lid + 4096, 0); }
            int pi;
            if (seen_epoch[lid] != epoch) {
                seen_epoch[lid] = epoch;
                pi = (int)p3d_ptr.size();
                seen_index[lid] = pi;
                p3d_opt.push_back(f3d->x); p3d_opt.push_back(f3d->y); p3d_opt.push_back(f3d->z);
                p3d_ptr.push_back(f3d.get());
            } else pi = seen_index[lid];
            obs_cam.push_back(ci); obs_pt.push_back(pi);
        }
    }
    const int n_obs = (int)obs_cam.size();
    // Only parameter blocks that appear in a residual block are part of the Ceres problem: compact the cameras.
    last_obs = (size_t)n_obs; last_points = p3d_ptr.size();
    std::vector<int> remap(cam_frame.size(), -1);
    std::vector<double> cams_c;
    cams_c.reserve(tr_opt.size());
    int nc = 0;
    {
        std::vector<uint8_t> used(cam_frame.size(), 0);
        for (int c : obs_cam) used[c] = 1;
        for (size_t c = 0; c < cam_frame.size(); c++)
            if (used[c]) { remap[c] = nc++; for (int k = 0; k < 6; k++) cams_c.push_back(tr_opt[c * 6 + k]); }
        for (int& c : obs_cam) c = remap[c];
    }
    delete hpg;
    tracker->stats.ba_calls++; tracker->stats.ba_obs += n_obs; tracker->stats.ba_points += (long)p3d_ptr.size();
    if (n_obs > 0) {
        const auto k0 = std::chrono::steady_clock::now();
        ba_solve(cams_c.data(), nc, p3d_opt.data(), (int)p3d_ptr.size(), obs.data(), obs_cam.data(), obs_pt.data(), n_obs,
                 tracker->camera, 1.0, tracker->cfg.ba_iterations);
        tracker->stats.t_ba_kernel += std::chrono::duration<double>(std::chrono::steady_clock::now() - k0).count();
    }
    HostProfScope hpsc(tracker->stats.hp.t[3]);
    for (size_t c = 0; c < cam_frame.size(); c++)
        if (remap[c] >= 0) for (int k = 0; k < 6; k++) tr_opt[c * 6 + k] = cams_c[remap[c] * 6 + k];
    // Updating 3D points and camera poses (:67-88)
    for (size_t c = 0; c < cam_frame.size(); c++) {
        const int i = cam_frame[c];
        const double rod[3] = {tr_opt[c * 6], tr_opt[c * 6 + 1], tr_opt[c * 6 + 2]};
        Mat3 _R;
        rodrigues_v2m(rod, _R.m);
        tracker->R[i] = _R.t();
        tracker->t[i] = Vec3{{-tr_opt[c * 6 + 3], -tr_opt[c * 6 + 4], -tr_opt[c * 6 + 5]}};
    }
    // (the reference repeats this loop once per window frame; Feature3D::update is a plain assignment, once is identical)
    if (!cam_frame.empty())
        for (size_t p = 0; p < p3d_ptr.size(); p++) p3d_ptr[p]->update(p3d_opt[p * 3], p3d_opt[p * 3 + 1], p3d_opt[p * 3 + 2]);
}

// ---- OdometryPipeline ------------------------------------------------------------------------------------------------
double OdometryPipeline::standardDeviation(const std::vector<double>& val) {   // :660-672
    double avg = 0, sd = 0;
    for (auto const& v : val) avg += v;
    avg /= val.size();
    for (auto const& v : val) sd += std::pow(v - avg, 2);
    return std::sqrt(sd / (val.size() - 1));
}

double OdometryPipeline::calcYRotation(const Mat3& R, bool flip) {   // OdometryPipeline.h:89-108
    const double c = R(0, 0), s = R(0, 2);
    if (flip) return s <= 0 ? -std::acos(c) : std::acos(c);
    return s <= 0 ? std::acos(c) : -std::acos(c);
}

std::vector<OdometryPipeline::GridSection> OdometryPipeline::getGridROI(Frame& fr) {   // :674-693
    std::vector<GridSection> roi;
    const int rows = fr.bw.h, cols = fr.bw.w;
    for (int r = 0; r < rows; r += cfg.grid_size[0])
        for (int c = 0; c < cols; c += cfg.grid_size[1]) {
            const int rw = std::min(cfg.grid_size[1], cols - c), rh = std::min(cfg.grid_size[0], rows - r);
            roi.push_back(GridSection{c / cfg.grid_size[1], r / cfg.grid_size[0], fr.regionOfInterest(c, r, rw, rh)});
        }
    return roi;
}

void OdometryPipeline::initialise() {   // :428-482
    int i = 0;
    Frame best = *(frames[0]);
    double cost = HUGE_VAL;
    for (auto& fr : frames) {
        std::vector<GridSection> roi = getGridROI(*fr);
        double n = cfg.min_tracked_features / roi.size();   // integer division (quirk Q6)
        std::vector<double> n_i, s_i;
        std::vector<Frame> cells;
        for (auto& r : roi) cells.push_back(r.frame);
        std::vector<std::vector<Feature>> all = extractor->extractGrid(cells, (int)n);
        stats.detect_calls++;
        for (size_t k = 0; k < roi.size(); k++) {
            std::vector<Feature>& feats = all[k];
            n_i.push_back((double)feats.size());
            for (auto& f : feats) {
                f.column = roi[k].x * cfg.grid_size[1] + f.column;
                f.row = roi[k].y * cfg.grid_size[0] + f.row;
                s_i.push_back(f.score);
                fr->map[std::make_shared<Feature>(f)] = std::weak_ptr<Feature3D>();
            }
        }
        const double std_n = standardDeviation(n_i), std_s = standardDeviation(s_i);
        const double _cost = std_n + std_s;
        if (_cost < cost) {
            fr->frame = 0;
            best = *fr;
            cost = _cost;
            init_offset = i;
        }
        i++;
    }
    frames.clear();
    frames.push_back(std::make_shared<Frame>(best));
}

void OdometryPipeline::addFrame(Frame& frame) {   // :329-374
    HostCpuScope cpu_(stats.hp.t[12]);
    frame.frame = (int)frames.size();
    const auto tl0 = std::chrono::steady_clock::now();
    fmap feat_corr = matcher->matchFeatures(*(frames[frame.frame - 1]), frame);
    stats.t_lk += std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count();
    stats.lk_calls++; stats.lk_points += (long)frames[frame.frame - 1]->map.size();
    const int n_corr = (int)feat_corr.size();
    frames[frame.frame - 1]->feat_corr = std::move(feat_corr);   // (the reference copies; the local is not used again)
    index_feat_corr(*frames[frame.frame - 1]);
    if (n_corr < cfg.tracked_features_tol) {
        std::vector<GridSection> roi = getGridROI(*(frames[frames.size() - 1]));   // cells of the PREVIOUS frame (quirk Q3)
        const int n_grid = (int)std::ceil((double)cfg.min_tracked_features / (double)roi.size());
        std::vector<Frame> cells;
        for (auto& r : roi) cells.push_back(r.frame);
        const auto td0 = std::chrono::steady_clock::now();
        std::vector<std::vector<Feature>> all = extractor->extractGrid(cells, n_grid);
        stats.t_detect += std::chrono::duration<double>(std::chrono::steady_clock::now() - td0).count();
        stats.detect_calls++;
        NeighborGrid near(frame);   // frame.hasNeighbor(f) for every candidate, without the scan per candidate
        for (size_t k = 0; k < roi.size(); k++)
            for (auto& f : all[k]) {
                if (!near.hasNeighbor(f.column, f.row)) {   // cell-LOCAL coordinates vs the global map (quirk Q4)
                    f.column = roi[k].x * cfg.grid_size[1] + f.column;
                    f.row = roi[k].y * cfg.grid_size[0] + f.row;
                    frame.map[std::make_shared<Feature>(f)] = std::weak_ptr<Feature3D>();
                    near.add(f.column, f.row);
                }
            }
    }
    frames.push_back(std::make_shared<Frame>(std::move(frame)));   // (the reference copies; callers only read frame.frame afterwards)
    // (no build_pnp_links here: in the one-thread schedule the lists would be built by the thread that then reads them - measured in the
    // batched leg, 128 such threads: +77 us of host CPU per frame for building against -15 us for the gathers)
}

void OdometryPipeline::motionHeuristics(Mat3& _R, Vec3& _t, int j) {   // :171-208
    if (_t(2) < 0 && calcYRotation(_R) < 3.1415 / 8 &&
        std::abs(_t(2)) > std::max(std::abs(_t(0)), std::abs(_t(1))) && std::abs(_t(2)) < 2 * scale) {
        t_s.push_back(_t);
        R_s.push_back(_R);
        _t = R[j] * _t + t[j];
        _R = _R * R[j];
    } else {
        stats.heuristic_motion++;
        t_s.push_back(t_s[j]);
        R_s.push_back(R_s[j]);
        _t = R[j] * t_s[j] + t[j];
        _R = R_s[j] * R[j];
    }
    t.push_back(_t);
    R.push_back(_R);
}

void OdometryPipeline::estimatePose(Frame& src, Frame& next) {   // :376-426
    HostProfScope hp_total(stats.hp.t[8]);
    HostCpuScope cpu_(stats.hp.t[13]);
    const int j = src.frame;
    Mat3 _R = R[j];
    Vec3 _t = t[j];
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
    int n3d;
    { HostProfScope h(stats.hp.t[7]); n3d = src.count3DPoints(); }
    const auto t0 = tnow();
    if (n3d >= cfg.tracked_features_tol) {
        pnpsolver->solvePnP(src, next, _R, _t);
        stats.t_pnp += secs(t0);
    } else {
        triangulator->triangulate(src, next, _R, _t);
        stats.tri_calls++;
        stats.t_tri += secs(t0);
    }
    { HostProfScope h(stats.hp.t[6]); motionHeuristics(_R, _t, j); }
    // (frames[src.frame], frames[next.frame] are updated in place; the reference works on copies and writes them back)
    if (cfg.bundle_size && src.frame && src.frame % (cfg.bundle_size / 3 * 2) == 0) {
        const auto t1 = tnow();
        HostCpuScope cpu_ba(stats.hp.t[15]);
        ba->apply(next);
        stats.t_ba += secs(t1);
    }
}

void OdometryPipeline::run() {   // startPipeline :247-264 + featureExtractionThread :212-229 + poseEstimationThread :237-243
    for (int i = 0; i < cfg.init_frames; i++) frames.push_back(std::make_shared<Frame>(Frame(images[i])));
    initialise();
    R.push_back(Mat3::eye()); t.push_back(Vec3{{0, 0, 0}});
    R_s.push_back(Mat3::eye()); t_s.push_back(Vec3{{0, 0, 0}});
    for (int i = init_offset + 1; i < (int)images.size(); i++) {
        if (i >= cfg.stop) break;
        Frame frame(images[i]);
        if (frame.isEmpty()) continue;
        addFrame(frame);
        if (on_frame_added) on_frame_added(frame.frame);
        if (frame.frame < 2) continue;
        const int j = frame.frame - 2;
        estimatePose(*frames[j], *frames[j + 1]);
    }
}

void OdometryPipeline::run_threaded() {
    for (int i = 0; i < cfg.init_frames; i++) frames.push_back(std::make_shared<Frame>(Frame(images[i])));
    { HostProfScope hp_init(stats.hp.t[11]); initialise(); }
    R.push_back(Mat3::eye()); t.push_back(Vec3{{0, 0, 0}});
    R_s.push_back(Mat3::eye()); t_s.push_back(Vec3{{0, 0, 0}});
    // job pipe (dlib::pipe<Job> in the reference). `frames` only grows at the back (front-end) while the back-end touches
    // entries j, j+1 <= k-1 that the front-end no longer reads; the vector itself is guarded by frames_mu, which
    // BundleAdjustmentBase::apply takes as well when it snapshots its window.
    std::mutex& mu = frames_mu;
    std::condition_variable cv, cv_space;
    std::deque<int> jobs;
    bool done = false;
    // dlib::pipe<Job> job_pipe(605) in the reference (OdometryPipeline.cpp:26): the front-end blocks when that many jobs wait. The depth
    // changes no result (the pipeline is schedule-deterministic); a short pipe keeps the frames the back-end is about to read in cache.
    const size_t pipe_depth = (size_t)std::max(1, cfg.pipe_depth);
    // a plugin error (e.g. a capacity error of the device library) in either thread ends the run and is rethrown to the caller
    std::exception_ptr back_error;
    std::atomic<bool> failed{false};
    std::thread back([&]() {
        try {
            for (;;) {
                int j;
                std::shared_ptr<Frame> a, b;
                {
                    HostProfScope hp_wait(stats.hp.t[9]);
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !jobs.empty() || done; });
                    if (jobs.empty()) return;
                    j = jobs.front(); jobs.pop_front();
                    a = frames[j]; b = frames[j + 1];
                    cv_space.notify_one();
                }
                estimatePose(*a, *b);
            }
        } catch (...) {
            back_error = std::current_exception();
            // under the pipe's mutex: the front-end evaluates `failed` in its wait predicate with the lock held, so the store cannot
            // fall between its test and its sleep (a notify without the lock could: the dead back-end never pops a job again)
            std::lock_guard<std::mutex> lk(mu);
            failed.store(true);
            cv_space.notify_all();
        }
    });
    std::exception_ptr front_error;
    const auto t_front0 = std::chrono::steady_clock::now();
    try {
    for (int i = init_offset + 1; i < (int)images.size(); i++) {
        if (i >= cfg.stop || failed.load()) break;
        Frame frame(images[i]);
        if (frame.isEmpty()) continue;
        {
            // addFrame reads frames[k-1] and appends frames[k]; BA snapshots tracker->frames[i] under the same mutex
            std::unique_lock<std::mutex> lk(mu);
            frame.frame = (int)frames.size();
        }
        std::shared_ptr<Frame> prev;
        { std::unique_lock<std::mutex> lk(mu); prev = frames[frame.frame - 1]; }
        const auto tl0 = std::chrono::steady_clock::now();
        fmap feat_corr = matcher->matchFeatures(*prev, frame);
        stats.t_lk += std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count();
        stats.lk_calls++; stats.lk_points += (long)prev->map.size();
        const int n_corr = (int)feat_corr.size();
        prev->feat_corr = std::move(feat_corr);
        index_feat_corr(*prev);
        if (triangulator) triangulator->prefetch(*prev);   // (prev, this frame) reaches the back-end one frame from now at the earliest
        if (n_corr < cfg.tracked_features_tol) {
            std::vector<GridSection> roi = getGridROI(*prev);
            const int n_grid = (int)std::ceil((double)cfg.min_tracked_features / (double)roi.size());
            std::vector<Frame> cells;
            for (auto& r : roi) cells.push_back(r.frame);
            const auto td0 = std::chrono::steady_clock::now();
            std::vector<std::vector<Feature>> all = extractor->extractGrid(cells, n_grid);
            stats.t_detect += std::chrono::duration<double>(std::chrono::steady_clock::now() - td0).count();
            stats.detect_calls++;
            NeighborGrid near(frame);
            for (size_t k = 0; k < roi.size(); k++)
                for (auto& f : all[k])
                    if (!near.hasNeighbor(f.column, f.row)) {
                        f.column = roi[k].x * cfg.grid_size[1] + f.column;
                        f.row = roi[k].y * cfg.grid_size[0] + f.row;
                        frame.map[std::make_shared<Feature>(f)] = std::weak_ptr<Feature3D>();
                        near.add(f.column, f.row);
                    }
        }
        const int frame_no = frame.frame;
        std::shared_ptr<Frame> stored = std::make_shared<Frame>(std::move(frame));   // built outside the lock
        build_pnp_links(*prev, *stored);   // the back-end reaches (prev, stored) after the next frame is in: nobody else touches either now
        {
            std::unique_lock<std::mutex> lk(mu);
            frames.push_back(std::move(stored));
            if (frame_no >= 2) {
                cv_space.wait(lk, [&] { return jobs.size() < pipe_depth || failed.load(); });   // job_pipe.enqueue blocks on a full pipe
                jobs.push_back(frame_no - 2); cv.notify_one();
            }
        }
        if (on_frame_added) on_frame_added(frame_no);
    }
    } catch (...) { front_error = std::current_exception(); }
    { std::unique_lock<std::mutex> lk(mu); done = true; cv.notify_one(); }
    stats.hp.t[10] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_front0).count();
    back.join();
    if (triangulator) triangulator->finish();   // helper threads of the two-view prefetch: none survives the run
    if (front_error) std::rethrow_exception(front_error);
    if (back_error) std::rethrow_exception(back_error);
}

}  // namespace vo
