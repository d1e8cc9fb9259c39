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

// ---- HashOrder: libstdc++'s _Hashtable (unique keys), replayed over node indices ----------------------------------------------
// Bucket counts of _Prime_rehash_policy as a function of the element count, read off the real container once (so the prime table
// and the growth rule are libstdc++'s own, whatever version this is built with).
size_t HashOrder::buckets_for(size_t n) {
    static const std::vector<unsigned>* table = [] {
        constexpr size_t N = 1u << 17;
        auto* t = new std::vector<unsigned>(N + 1);
        std::unordered_map<int, int> m;
        (*t)[0] = (unsigned)m.bucket_count();
        for (size_t i = 1; i <= N; i++) { m.emplace((int)i, 0); (*t)[i] = (unsigned)m.bucket_count(); }
        return t;
    }();
    if (n < table->size()) return (*table)[n];
    std::unordered_map<int, int> m;   // (beyond the table: ask the container itself)
    for (size_t i = 1; i <= n; i++) m.emplace((int)i, 0);
    return m.bucket_count();
}
// _M_rehash_aux(n, true_type): the nodes are re-linked in their current order; a node whose new bucket is still empty goes to the
// FRONT of the list, otherwise behind the first node ("before" node) of its bucket
void HashOrder::rehash(size_t n) {
    std::vector<int>& nb = bscratch_;
    nb.assign(n, EMPTY);
    int p = head_;
    head_ = -1;
    size_t bbegin_bkt = 0;
    while (p >= 0) {
        const int nx = next_[(size_t)p];
        const size_t bkt = code_[(size_t)p] % n;
        if (nb[bkt] == EMPTY) {
            next_[(size_t)p] = head_;
            head_ = p;
            nb[bkt] = BEFORE_BEGIN;
            if (next_[(size_t)p] >= 0) nb[bbegin_bkt] = p;
            bbegin_bkt = bkt;
        } else if (nb[bkt] == BEFORE_BEGIN) {
            next_[(size_t)p] = head_;
            head_ = p;
        } else {
            const int prev = nb[bkt];
            next_[(size_t)p] = next_[(size_t)prev];
            next_[(size_t)prev] = p;
        }
        p = nx;
    }
    bprev_.swap(nb);
    nb_ = n;
}
// _M_insert_unique_node: grow first (by the policy's count for one more element), then _M_insert_bucket_begin
int HashOrder::insert(size_t code) {
    const int node = (int)next_.size();
    const size_t want = buckets_for((size_t)node + 1);
    if (want != nb_ || bprev_.empty()) rehash(want);
    const size_t bkt = code % nb_;
    next_.push_back(-1);
    code_.push_back(code);
    const int prev = bprev_[bkt];
    if (prev == BEFORE_BEGIN) { next_[(size_t)node] = head_; head_ = node; }
    else if (prev != EMPTY) { next_[(size_t)node] = next_[(size_t)prev]; next_[(size_t)prev] = node; }
    else {
        next_[(size_t)node] = head_;
        head_ = node;
        if (next_[(size_t)node] >= 0) bprev_[code_[(size_t)next_[(size_t)node]] % nb_] = node;
        bprev_[bkt] = BEFORE_BEGIN;
    }
    return node;
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
    std::vector<Feature> new_feats;
    std::vector<int> old_feats;   // features of src in the order src.map is walked
    std::vector<Feature> cmp_feats = extractor->extractFeatures(next, 1000);   // :11, on the whole next frame
    src.for_each_feature([&](int e) { old_feats.push_back(e); });
    const int n = (int)old_feats.size(), m = (int)cmp_feats.size();
    std::vector<int> src_xy(2 * (size_t)n), cmp_xy(2 * (size_t)m), best(n, -1);
    std::vector<float> errs(n, 0.f);
    for (int i = 0; i < n; i++) { src_xy[2 * i] = src.column[(size_t)old_feats[i]]; src_xy[2 * i + 1] = src.row[(size_t)old_feats[i]]; }
    for (int j = 0; j < m; j++) { cmp_xy[2 * j] = cmp_feats[j].column; cmp_xy[2 * j + 1] = cmp_feats[j].row; }
    if (n > 0) knn(src.bw, next.bw, src_xy.data(), n, cmp_xy.data(), m, best.data(), errs.data());   // :15-31 for every feature
    for (int i = 0; i < n; i++) {
        Feature best_fit = best[i] >= 0 ? cmp_feats[best[i]] : Feature();   // the neighbour copy (default-constructed when there was none)
        const float err = errs[i];
        if (err < threshold) {
            best_fit.tracked = true;
            best_fit.displacement = Feature(src_xy[2 * i], src_xy[2 * i + 1]).distance(best_fit);
            avg += best_fit.displacement;
        }   // (else: the copy keeps the candidate's own `tracked` flag: quirk of the reference, :32-41)
        new_feats.push_back(best_fit);
    }
    avg /= (double)new_feats.size();
    for (auto& f : new_feats)
        if (f.displacement > 3 * avg) f.tracked = false;
    for (size_t i = 0; i < old_feats.size(); i++)
        if (new_feats[i].tracked) {
            const int e_new = next.add_feature(new_feats[i].column, new_feats[i].row, src.lm[(size_t)old_feats[i]]);   // next.map[new] = src.map[old]
            const int c = src.corr_at(map, old_feats[i]);                                                            // map[old] = new
            map.val[(size_t)c] = e_new;
        }
    return map;
}

// ---- OpenCVLucasKanadeFM.cpp:5-32 ----------------------------------------------------------------------------
fmap LucasKanadeFMBase::matchFeatures(Frame& src, Frame& next) {
    fmap correspondences;
    const int n = src.n_features();
    std::vector<int> walk((size_t)n);           // src.map in iteration order
    std::vector<float> prev_points(2 * (size_t)n), next_points(2 * (size_t)n);
    {
        int i = 0;
        src.for_each_feature([&](int e) {
            walk[(size_t)i] = e;
            prev_points[2 * (size_t)i] = (float)src.column[(size_t)e];
            prev_points[2 * (size_t)i + 1] = (float)src.row[(size_t)e];
            i++;
        });
    }
    std::vector<uint8_t> status((size_t)n);
    std::vector<float> err((size_t)n);
    if (n > 0) pyrlk(src.bw, next.bw, prev_points.data(), n, next_points.data(), status.data(), err.data());
    next.column.reserve((size_t)n + 512); next.row.reserve((size_t)n + 512); next.lm.reserve((size_t)n + 512); next.map_order.reserve_nodes((size_t)n + 512);
    correspondences.key.reserve((size_t)n); correspondences.val.reserve((size_t)n); correspondences.order.reserve_nodes((size_t)n);
    for (int i = 0; i < n; i++) {
        if (!status[(size_t)i]) continue;
        const int e = walk[(size_t)i];
        // Feature(next_points[i].x, next_points[i].y): float -> int truncation toward zero
        const int f = next.add_feature((int)next_points[2 * (size_t)i], (int)next_points[2 * (size_t)i + 1], src.lm[(size_t)e]);   // next.map[f] = p.second
        const int c = src.corr_at(correspondences, e);   // correspondences[p.first] = f (same-pixel sources share the entry; the last value wins)
        correspondences.val[(size_t)c] = f;
    }
    return correspondences;
}

// ---- OpenCVEPnPSolver.cpp:4-50 ---------------------------------------------------------------------------------
void EPnPSolverBase::solvePnP(Frame& src, Frame& next, Mat3& R_out, Vec3& t_out) {
    const int j = src.frame;
    std::vector<float> obj_points, img_points;
    double _R_rod[3];
    rodrigues_m2v(R_out.m, _R_rod);
    std::vector<int> local_feats3d;
    LandmarkTable& L = tracker->landmarks;
    HostProfScope* hps = new HostProfScope(tracker->stats.hp.t[0]);
    obj_points.reserve(3 * (size_t)src.n_features()); img_points.reserve(2 * (size_t)src.n_features()); local_feats3d.reserve((size_t)src.n_features());
    const Mat3& Rj = tracker->R[j];
    const Vec3& tj = tracker->t[j];
    src.for_each_feature([&](int e) {   // for (auto& p : src.map)
        const int id = src.lm[(size_t)e];
        if (L.expired(id)) return;
        const int c = src.corr_at(src.feat_corr, e);   // src.feat_corr[p.first]: operator[] inserts an empty entry when there is none (quirk Q10)
        const int f = src.feat_corr.val[(size_t)c];
        if (f < 0) return;                             // .expired()
        next.lm[(size_t)f] = id;                       // next.map[f] = weak_ptr(f3d)
        Feature3D f3d = L.get(id);
        f3d.transformInv(Rj, tj);
        float px = f3d.x, py = f3d.y, pz = f3d.z;
        pz *= -1;
        obj_points.push_back(px); obj_points.push_back(py); obj_points.push_back(pz);
        img_points.push_back((float)next.column[(size_t)f]); img_points.push_back((float)next.row[(size_t)f]);
        f3d.transform(Rj, tj);                         // float round trip in place (quirk Q7)
        L.put(id, f3d);
        local_feats3d.push_back(id);
    });
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
    std::vector<uint8_t> is_inlier((size_t)m, 0);
    for (int i : inliers) if (i >= 0 && i < m) is_inlier[(size_t)i] = 1;
    for (int i = 0; i < m; i++)
        if (!is_inlier[(size_t)i] && !L.expired(local_feats3d[(size_t)i])) L.erase(local_feats3d[(size_t)i]);
}

// ---- CeresBundleAdjustment.cpp:5-89 ------------------------------------------------------------------------------
void BundleAdjustmentBase::apply(Frame& f) {
    const int fn = (int)f.frame + 1;
    const int n = std::min(tracker->cfg.bundle_size, fn);
    LandmarkTable& L = tracker->landmarks;
    // Work vectors: local, sized from the previous solve so that they do not regrow element by element.
    std::vector<int> cam_frame, obs_cam, obs_pt;      // window frames in order (skipping 0); camera / point index per residual block
    std::vector<double> tr_opt, obs;                  // 6 per window frame; 2 per residual block
    cam_frame.reserve((size_t)n); tr_opt.reserve((size_t)6 * n);
    obs.reserve(2 * last_obs + 64); obs_cam.reserve(last_obs + 32); obs_pt.reserve(last_obs + 32);
    const unsigned epoch = ++epoch_counter;   // p3d_opt.count(f3d) of the reference, as epoch-stamped arrays over the landmark ids
    std::vector<int> p3d_id;
    std::vector<double> p3d_opt;
    p3d_id.reserve(last_points + 32); p3d_opt.reserve(3 * last_points + 96);
    HostProfScope* hpg = new HostProfScope(tracker->stats.hp.t[2]);
    std::vector<std::shared_ptr<Frame>> window((size_t)n);   // snapshot under the lock: the front-end thread may be appending
    {
        std::lock_guard<std::mutex> lk(tracker->frames_mu);
        for (int i = fn - n; i < fn; i++) window[(size_t)(i - (fn - n))] = tracker->frames[i];
    }
    if (seen_epoch.size() < (size_t)L.size()) { seen_epoch.resize((size_t)L.size() + 4096, 0); seen_index.resize((size_t)L.size() + 4096, 0); }
    for (int i = fn - n; i < fn; i++) {
        if (i == 0) continue;
        const Frame& frame = *window[(size_t)(i - (fn - n))];
        double rod[3];
        Mat3 Rt = tracker->R[i].t();
        rodrigues_m2v(Rt.m, rod);
        const int ci = (int)cam_frame.size();
        cam_frame.push_back(i);
        tr_opt.push_back(rod[0]); tr_opt.push_back(rod[1]); tr_opt.push_back(rod[2]);
        tr_opt.push_back(-tracker->t[i].v[0]); tr_opt.push_back(-tracker->t[i].v[1]); tr_opt.push_back(-tracker->t[i].v[2]);
        frame.for_each_feature([&](int e) {   // for (auto& p : frame->map)
            const int id = frame.lm[(size_t)e];
            if (L.expired(id)) return;
            obs.push_back((double)frame.column[(size_t)e]); obs.push_back((double)frame.row[(size_t)e]);
            // index of the landmark in first-seen order (the parameter blocks of the reference's problem, in the order they enter it)
            int pi;
            if (seen_epoch[(size_t)id] != epoch) {
                seen_epoch[(size_t)id] = epoch;
                pi = (int)p3d_id.size();
                seen_index[(size_t)id] = pi;
                p3d_opt.push_back(L.xyz[3 * (size_t)id]); p3d_opt.push_back(L.xyz[3 * (size_t)id + 1]); p3d_opt.push_back(L.xyz[3 * (size_t)id + 2]);
                p3d_id.push_back(id);
            } else pi = seen_index[(size_t)id];
            obs_cam.push_back(ci); obs_pt.push_back(pi);
        });
    }
    const int n_obs = (int)obs_cam.size();
    // Only parameter blocks that appear in a residual block are part of the Ceres problem: compact the cameras.
    last_obs = (size_t)n_obs; last_points = p3d_id.size();
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
    tracker->stats.ba_calls++; tracker->stats.ba_obs += n_obs; tracker->stats.ba_points += (long)p3d_id.size();
    if (n_obs > 0) {
        const auto k0 = std::chrono::steady_clock::now();
        ba_solve(cams_c.data(), nc, p3d_opt.data(), (int)p3d_id.size(), obs.data(), obs_cam.data(), obs_pt.data(), n_obs,
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
        for (size_t p = 0; p < p3d_id.size(); p++) L.put(p3d_id[p], Feature3D(p3d_opt[p * 3], p3d_opt[p * 3 + 1], p3d_opt[p * 3 + 2]));
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
                fr->add_feature(f.column, f.row, -1);   // fr->map[make_shared<Feature>(f)] = weak_ptr<Feature3D>()
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
    Frame& prev = *(frames[frame.frame - 1]);
    const auto tl0 = std::chrono::steady_clock::now();
    fmap feat_corr = matcher->matchFeatures(prev, frame);
    stats.t_lk += std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count();
    stats.lk_calls++; stats.lk_points += (long)prev.n_features();
    const int n_corr = (int)feat_corr.size();
    prev.feat_corr = std::move(feat_corr);   // (the reference copies; the local is not used again)
    redetect(prev, frame, n_corr);
    frames.push_back(std::make_shared<Frame>(std::move(frame)));   // (the reference copies; callers only read frame.frame afterwards)
}

// the second half of addFrame (:342-371): new features when too few were tracked
void OdometryPipeline::redetect(Frame& prev, Frame& frame, int n_corr) {
    if (n_corr >= cfg.tracked_features_tol) return;
    std::vector<GridSection> roi = getGridROI(prev);   // cells of the PREVIOUS frame (quirk Q3)
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
                frame.add_feature(f.column, f.row, -1);   // frame.map[make_shared<Feature>(f)] = weak_ptr<Feature3D>()
                near.add(f.column, f.row);
            }
        }
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
    { HostProfScope h(stats.hp.t[7]); n3d = src.count3DPoints(landmarks); }
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

// ---- retired frames ------------------------------------------------------------------------------------------------------
void OdometryPipeline::retire_before(int k_end) {
    constexpr size_t CHUNK = 1u << 18;   // ints per store chunk
    for (; retired_upto < k_end; retired_upto++) {
        const int k = retired_upto;
        std::shared_ptr<Frame> f;
        {
            std::lock_guard<std::mutex> lk(frames_mu);
            if ((size_t)k >= frames.size() || !frames[(size_t)k] || frames[(size_t)k].use_count() != 1) return;   // still shared (a BA snapshot): next time
            f = std::move(frames[(size_t)k]);
            frames[(size_t)k].reset();
            if (retired.size() < frames.size()) retired.resize(frames.size() + 1024);
        }
        RetiredFrame& r = retired[(size_t)k];
        const size_t need = 3 * (size_t)f->n_features();
        if (retired_store.empty() || retired_store.back().size() + need > retired_store.back().capacity()) {
            retired_store.emplace_back();
            retired_store.back().reserve(std::max(CHUNK, need));
        }
        std::vector<int>& st = retired_store.back();
        r.chunk = retired_store.size() - 1; r.offset = st.size();
        f->for_each_feature([&](int e) { st.push_back(f->column[(size_t)e]); st.push_back(f->row[(size_t)e]); st.push_back(f->lm[(size_t)e]); });
        r.n_corr = (int)f->feat_corr.size();
        r.n_features = f->n_features();
        std::lock_guard<std::mutex> lk(frames_mu);
        spare.push_back(std::move(*f));
    }
}

Frame OdometryPipeline::take_frame(const ImageView& img) {
    {
        std::lock_guard<std::mutex> lk(frames_mu);
        if (!spare.empty()) {
            Frame f = std::move(spare.back());
            spare.pop_back();
            f.reset(img);
            return f;
        }
    }
    return Frame(img);
}

void OdometryPipeline::frame_features(int k, int* out) const {
    if (frames[(size_t)k]) {
        const Frame& fr = *frames[(size_t)k];
        int i = 0;
        fr.for_each_feature([&](int e) {
            out[3 * i] = fr.column[(size_t)e]; out[3 * i + 1] = fr.row[(size_t)e];
            out[3 * i + 2] = landmarks.expired(fr.lm[(size_t)e]) ? -1 : fr.lm[(size_t)e];
            i++;
        });
        return;
    }
    const RetiredFrame& r = retired[(size_t)k];
    const int* src = retired_store[r.chunk].data() + r.offset;
    for (int i = 0; i < r.n_features; i++) {
        out[3 * i] = src[3 * i]; out[3 * i + 1] = src[3 * i + 1];
        out[3 * i + 2] = landmarks.expired(src[3 * i + 2]) ? -1 : src[3 * i + 2];
    }
}

void OdometryPipeline::run() {   // startPipeline :247-264 + featureExtractionThread :212-229 + poseEstimationThread :237-243
    for (int i = 0; i < cfg.init_frames; i++) frames.push_back(std::make_shared<Frame>(Frame(images[i])));
    initialise();
    R.push_back(Mat3::eye()); t.push_back(Vec3{{0, 0, 0}});
    R_s.push_back(Mat3::eye()); t_s.push_back(Vec3{{0, 0, 0}});
    for (int i = init_offset + 1; i < (int)images.size(); i++) {
        if (i >= cfg.stop) break;
        if (images[i].w == 0) continue;   // (Frame::isEmpty)
        Frame frame = take_frame(images[i]);
        addFrame(frame);
        if (on_frame_added) on_frame_added(frame.frame);
        if (frame.frame < 2) continue;
        const int j = frame.frame - 2;
        estimatePose(*frames[j], *frames[j + 1]);
        retire_before(j + 1 - std::max(cfg.bundle_size, 2));   // the next job's bundle window starts at j + 3 - bundle_size
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
                a.reset(); b.reset();
                retire_before(j + 1 - std::max(cfg.bundle_size, 2));   // the next job's bundle window starts at j + 3 - bundle_size
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
        if (images[i].w == 0) continue;   // (Frame::isEmpty)
        Frame frame = take_frame(images[i]);
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
        stats.lk_calls++; stats.lk_points += (long)prev->n_features();
        const int n_corr = (int)feat_corr.size();
        prev->feat_corr = std::move(feat_corr);
        if (triangulator) triangulator->prefetch(*prev, frame);   // (prev, this frame) reaches the back-end one frame from now at the earliest
        redetect(*prev, frame, n_corr);
        const int frame_no = frame.frame;
        std::shared_ptr<Frame> stored = std::make_shared<Frame>(std::move(frame));   // built outside the lock
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
