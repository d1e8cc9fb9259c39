// Host orchestration of the VO hot path: an OpenCV/dlib-free mirror of the reference's OdometryPipeline
// (/root/reference/OdometryPipeline.cpp: initialise :428-482, addFrame :329-374, estimatePose :376-426,
// motionHeuristics :171-208, getGridROI :674-693, standardDeviation :660-672) plus the gather/scatter halves of its
// plugin adapters (OpenCVLucasKanadeFM.cpp:5-32, OpenCVGoodFeatureExtractor.cpp:4-21, OpenCVEPnPSolver.cpp:4-50,
// OpenCVFivePointTri.cpp:5-54, CeresBundleAdjustment.cpp:5-89).  The numerics those adapters hand to OpenCV/Ceres are
// virtual "kernel" hooks: the product implements them with the HIP C ABI (include/pmv_hip.h), the oracle with its CPU
// restatement.  GUI, drawing, video, config parsing and the error file are out of scope.
#pragma once
#include <exception>
#include <unordered_map>
#include <thread>
#include <deque>
#include <condition_variable>
#include <atomic>
#include "vo_types.h"
#include <functional>
#include <chrono>
#include <mutex>
#include <time.h>

namespace vo {

struct Config {   // the keys of the reference's config file that reach the hot path (OdometryPipeline.cpp:50-58)
    int min_tracked_features = 400;
    int tracked_features_tol = 150;
    int init_frames = 5;
    int stop = 1 << 30;       // "frames"
    int bundle_size = 5;
    int ba_iterations = 5;    // "max_iterations"
    int grid_size[2] = {255, 255};   // OdometryPipeline.h:31
    // plugin constants hard-coded in the reference
    double gftt_quality = 0.01, gftt_min_distance = 5;   // OpenCVGoodFeatureExtractor.h:9,11
    int extractor = 0;        // 0 = OpenCVGoodFeatureExtractor (default, OdometryPipeline.cpp:68), 1 = ShiTomasiFeatureExtractor
    double shitomasi_quality = 0.4;                      // ShiTomasiFeatureExtractor.h:10
    int pipe_depth = 605;     // jobs the front-end may be ahead of the back-end: dlib::pipe<Job> job_pipe(605), OdometryPipeline.cpp:26
};

class OdometryPipeline;

// ---- adapters: the reference's plugin classes with the third-party call factored into a pure-virtual hook ----
class GoodFeatureExtractorBase : public BaseFeatureExtractor {   // OpenCVGoodFeatureExtractor
public:
    double quality = 0.01, min_distance = 5;
    // cells share one full image; out[i] = corners (x,y) of cell i in cell coordinates, OpenCV order
    virtual void gftt(const std::vector<ImageView>& cells, int max, std::vector<std::vector<std::pair<int, int>>>& out) = 0;
    std::vector<Feature> extractFeatures(Frame& src, int max) override;
    std::vector<std::vector<Feature>> extractGrid(std::vector<Frame>& cells, int max) override;
};
class ShiTomasiExtractorBase : public BaseFeatureExtractor {     // ShiTomasiFeatureExtractor
public:
    double quality = 0.4;
    virtual void shitomasi(const std::vector<ImageView>& cells, int max, std::vector<std::vector<std::pair<int, int>>>& out,
                           std::vector<std::vector<double>>& score) = 0;
    std::vector<Feature> extractFeatures(Frame& src, int max) override;
    std::vector<std::vector<Feature>> extractGrid(std::vector<Frame>& cells, int max) override;
};
class FastExtractorBase : public BaseFeatureExtractor {          // OpenCVFASTFeatureExtractor (threshold 10, nonmax: OpenCVFASTFeatureExtractor.h:11-12)
public:
    int threshold = 10;
    bool nonmax = true;
    // cv::FAST on each view; out[i] = the first `max` keypoints (x, y) of view i in cv::FAST's order, response[i] their KeyPoint::response
    virtual void fast(const std::vector<ImageView>& cells, int max, std::vector<std::vector<std::pair<int, int>>>& out,
                      std::vector<std::vector<float>>& response) = 0;
    std::vector<Feature> extractFeatures(Frame& src, int max) override;
    std::vector<std::vector<Feature>> extractGrid(std::vector<Frame>& cells, int max) override;
};
class KnnFeatureMatcherBase : public BaseFeatureMatcher {        // kNNFeatureMatcher (window 15, threshold 2, 7 neighbours: kNNFeatureMatcher.h:11-12,31)
public:
    int window = 15, threshold = 2, neighbours = 7;
    BaseFeatureExtractor* extractor = nullptr;                   // kNNFeatureMatcher(BaseFeatureExtractor*): called on the whole next frame
    // getNearestNeighbors + compareFeatures + best-fit rule for n source features against m candidates (pixel coordinates in the
    // full images of src / next): best[i] = candidate index or -1 (the default Feature at (0,0)), err[i] = its window error
    virtual void knn(const ImageView& src, const ImageView& next, const int* src_xy, int n, const int* cmp_xy, int m, int* best, float* err) = 0;
    fmap matchFeatures(Frame& src, Frame& next) override;
};
class LucasKanadeFMBase : public BaseFeatureMatcher {            // OpenCVLucasKanadeFM (win 32, 4 levels)
public:
    virtual void pyrlk(const ImageView& prev, const ImageView& next, const float* prev_xy, int n, float* next_xy,
                       uint8_t* status, float* err) = 0;
    fmap matchFeatures(Frame& src, Frame& next) override;
};
class alignas(64) EPnPSolverBase : public BasePnPSolver {                    // OpenCVEPnPSolver
public:
    OdometryPipeline* tracker = nullptr;
    // cv::solvePnPRansac(obj, img, K, noDist, rvec, tvec, true, 100, 8, .99, inliers); returns false on failure
    virtual bool pnp_ransac(const float* obj_xyz, const float* img_xy, int m, const double* K, double* rvec, double* tvec,
                            std::vector<int>& inliers) = 0;
    void solvePnP(Frame& src, Frame& next, Mat3& R, Vec3& t) override;
};
// The per-point part of cv::recoverPose: for each of the four (R, t) candidates of an essential matrix, DLT-triangulate
// every correspondence (cvTriangulatePoints: eigenvector of the smallest eigenvalue of A^T A, 4x4) and apply the cheirality
// tests. q1/q2: n normalised image points (x, y); P1x4: four 3x4 camera matrices [R | t]; mask_in: n bytes (RANSAC inliers);
// out_Q: [candidate][4][n] homogeneous points; out_mask: [candidate][n]; out_good: [4] counts. Host loops (oracle + default).
void dlt_candidates_host(const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in, double* out_Q,
                         uint8_t* out_mask, int* out_good);

class SpinPool;
class alignas(64) FivePointTri : public BaseTriangulator {                   // OpenCVFivePointTri
public:
    OdometryPipeline* tracker = nullptr;
    int workers = 1;                 // threads evaluating RANSAC hypotheses side by side (results do not depend on it)
    std::shared_ptr<SpinPool> pool;  // created on first use when workers > 1
    void triangulate(Frame& src, Frame& next, Mat3& R, Vec3& t) override;
    // findEssentialMat ahead of time: E of a frame pair depends on the 2-D correspondences only (OpenCVFivePointTri.cpp:8-24), which
    // the front-end has a frame before the back-end asks. With prefetch_threads > 0 the front-end hands each pair's points to
    // that many helper threads; triangulate() takes the finished result (or claims the job and computes it itself when no helper
    // has started it). Same function on the same points: identical E, mask and iteration count. Off for the hook form.
    int prefetch_threads = 0;
    void prefetch(const Frame& prev, const Frame& next) override;
    void finish() override;   // stops and joins the helper threads (jobs nobody asked for are dropped): nothing of this run touches the plugin hooks afterwards
    ~FivePointTri() override;
    // optional kernel hook for the RANSAC hypotheses of findEssentialMat: for n_hyp samples (5 indices each) of the n normalised
    // correspondences return the essential matrices of every sample (models: n_hyp x 90, n_models: n_hyp) and their inlier counts
    // under the float32 Sampson test (counts: n_hyp x 10). Return false (default) to evaluate them on host threads instead.
    static constexpr int HYP_ROUND = 32;   // samples handed to the hook per round
    bool use_hypothesis_hook = false;      // set by plugins that implement essential_hypotheses
    virtual bool essential_hypotheses(const double* q1, const double* q2, int n, const int* samples, int n_hyp, float thr, double* models,
                                      int* n_models, int* counts) {
        (void)q1; (void)q2; (void)n; (void)samples; (void)n_hyp; (void)thr; (void)models; (void)n_models; (void)counts;
        return false;
    }
    // kernel hook (same contract as dlt_candidates_host); the HIP plugin overrides it with pmv_triangulate_candidates
    virtual void dlt_candidates(const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in, double* out_Q,
                                uint8_t* out_mask, int* out_good) {
        dlt_candidates_host(q1, q2, n, P1x4, mask_in, out_Q, out_mask, out_good);
    }
    // the same kernel hook when called from a prefetch helper thread, concurrently with the back-end thread's own plugin calls: a plugin
    // whose dlt_candidates is not re-entrant overrides it (the HIP plugin: the context's auxiliary lane)
    virtual void dlt_candidates_ahead(const double* q1, const double* q2, int n, const double* P1x4, const uint8_t* mask_in, double* out_Q,
                                      uint8_t* out_mask, int* out_good) {
        dlt_candidates_host(q1, q2, n, P1x4, mask_in, out_Q, out_mask, out_good);
    }
    long prefetch_hits = 0, prefetch_inline = 0;   // triangulate() calls served by a helper / computed by the caller
private:
    struct EssentialJob {
        int frame = 0;
        std::vector<double> p1, p2;
        double E[9] = {0};
        std::vector<uint8_t> mask;
        int drawn = 0;
        bool ok = false;
        // cv::recoverPose of the pair (it needs E and the points only): rotation, unit translation, updated mask, triangulated points
        double R[9] = {0}, t[3] = {0};
        std::vector<double> tri;
        std::exception_ptr error;    // a plugin error in the helper is rethrown by triangulate() on the back-end thread
        std::atomic<int> state{0};   // 0 queued, 1 claimed (helper or caller), 2 done
    };
    std::mutex pf_mu;
    std::condition_variable pf_cv;
    std::deque<std::shared_ptr<EssentialJob>> pf_queue;
    std::unordered_map<int, std::shared_ptr<EssentialJob>> pf_jobs;
    std::vector<std::thread> pf_threads;
    bool pf_stop = false;
    void prefetch_worker();
};
class alignas(64) BundleAdjustmentBase : public BaseOptimizer {              // CeresBundleAdjustment
public:
    OdometryPipeline* tracker = nullptr;
    // ceres::Solve on cams (nc x 6: [aa(R^T), -t]) and pts (np x 3); in place
    virtual void ba_solve(double* cams, int nc, double* pts, int np, const double* obs_xy, const int* cam_idx,
                          const int* pt_idx, int n_obs, const double* K, double huber, int max_iterations) = 0;
    void apply(Frame& src) override;
protected:
    std::vector<unsigned> seen_epoch;   // per landmark id: the apply() call that last saw it / its index in that call
    std::vector<int> seen_index;
    unsigned epoch_counter = 0;
    size_t last_obs = 0, last_points = 0;   // sizes of the previous solve: reserve() for the next
};

// optional section timers of the host adapters (PMV_HOST_PROF=1 prints them to stderr at the end of a run)
struct HostProf {
    static constexpr int N = 16;
    double t[N] = {0};
    static const char* name(int i) {
        static const char* n[N] = {"pnp_gather", "pnp_scatter", "ba_gather", "ba_scatter", "tri_gather", "tri_landmarks", "heuristics",
                                   "count3d", "estimatePose", "backend_wait", "frontend_total", "init",
                                   "cpu_addFrame", "cpu_estimatePose", "cpu_fivepoint", "cpu_ba_apply"};   // thread CPU seconds (not wall)
        return n[i];
    }
};
struct HostCpuScope {   // CPU time of the calling thread spent inside the scope (blocked waits do not count)
    double& acc; double t0;
    static double now() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
    explicit HostCpuScope(double& a) : acc(a), t0(now()) {}
    ~HostCpuScope() { acc += now() - t0; }
};
struct HostProfScope {
    double& acc; std::chrono::steady_clock::time_point t0;
    explicit HostProfScope(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~HostProfScope() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// per-call statistics for the bench (not part of any result)
struct Stats {
    long lk_calls = 0, lk_points = 0, detect_calls = 0, pnp_calls = 0, pnp_points = 0, tri_calls = 0, ba_calls = 0, ba_obs = 0, ba_points = 0;
    long heuristic_motion = 0;
    // wall time per stage as seen by the calling host thread (adapter gather/scatter + plugin kernel + sync)
    double t_lk = 0, t_detect = 0, t_pnp = 0, t_tri = 0, t_ba = 0, t_pnp_kernel = 0, t_ba_kernel = 0;
    double t_tri_essential = 0, t_tri_pose = 0, tri_hypotheses = 0;
    long tri_ahead = 0;   // triangulate() calls whose two-view geometry had been computed ahead by a prefetch helper
    HostProf hp;   // inside t_tri: five-point RANSAC, recoverPose; RANSAC samples drawn
};

class OdometryPipeline {
public:
    Config cfg;
    double scale = 1;
    int init_offset = 0;
    double camera[9];                       // row-major 3x3
    std::vector<ImageView> images;          // the sequence ("file_names"): decoded gray frames
    std::vector<Vec3> gt_t;                 // ground-truth positions (parsePoses), used for scale only (Q11)
    // feats3d of the reference (a vector of shared_ptr<Feature3D>, RANSAC outliers erased with std::find + erase,
    // OpenCVEPnPSolver.cpp:47) as a table: id = creation number, liveness = "still in feats3d"
    LandmarkTable landmarks;
    std::vector<std::shared_ptr<Frame>> frames;
    // guards the `frames` VECTOR (push_back by the front-end thread, element reads by the back-end thread: estimatePose's job
    // hand-over and BundleAdjustmentBase::apply's window). The Frame objects themselves are never touched by both threads at
    // once. frame_mutex of the reference (OdometryPipeline.h) plays the same role.
    std::mutex frames_mu;
    // Frames nothing reads any more (older than the bundle window of the job the back-end has finished) are RETIRED: what the exports
    // need of them - the (column, row, landmark id) triples in the map's iteration order and the size of feat_corr - moves into
    // `retired`, frames[k] becomes null and the frame's tables (their capacity) go to `spare` for the frames to come. The reference keeps
    // every Frame alive until the end (~70 KB each: 75 MB per KITTI-00-length sequence); with 192 sequences per process that was 3 GB/s
    // of fresh heap, and a sixth of the host CPU time of a batched run sat in mprotect / page faults (scripts/hostprof).
    struct RetiredFrame { int n_features = -1, n_corr = 0; size_t chunk = 0, offset = 0; };
    std::vector<RetiredFrame> retired;            // per frame number; n_features < 0 = not retired (then frames[k] is alive)
    std::vector<std::vector<int>> retired_store;  // the triples, in chunks (one allocation per ~1 MB instead of one per frame)
    std::vector<Frame> spare;                     // guarded by frames_mu
    int retired_upto = 0;
    void retire_before(int k_end);                // called by the thread that runs estimatePose, after a job
    Frame take_frame(const ImageView& img);       // a recycled Frame(img), or a new one
    int frame_feature_count(int k) const { return frames[(size_t)k] ? frames[(size_t)k]->n_features() : retired[(size_t)k].n_features; }
    int frame_corr_count(int k) const { return frames[(size_t)k] ? (int)frames[(size_t)k]->feat_corr.size() : retired[(size_t)k].n_corr; }
    void frame_features(int k, int* out3) const;  // (column, row, landmark id or -1 when the landmark has expired) in iteration order
    std::vector<Mat3> R, R_s;
    std::vector<Vec3> t, t_s;
    BaseFeatureExtractor* extractor = nullptr;
    BaseFeatureMatcher* matcher = nullptr;
    BasePnPSolver* pnpsolver = nullptr;
    BaseTriangulator* triangulator = nullptr;
    BaseOptimizer* ba = nullptr;
    Stats stats;

    struct GridSection { int x, y; Frame frame; };

    void initialise();                                  // :428-482
    void addFrame(Frame& frame);                        // :329-374
    void redetect(Frame& prev, Frame& frame, int n_corr);   // its second half (:342-371), shared by the two schedules
    void estimatePose(Frame& src, Frame& next);         // :376-426
    void motionHeuristics(Mat3& _R, Vec3& _t, int j);   // :171-208
    std::vector<GridSection> getGridROI(Frame& fr);     // :674-693
    static double standardDeviation(const std::vector<double>& val);   // :660-672
    static double calcYRotation(const Mat3& R, bool flip = false);     // OdometryPipeline.h:89-108
    // startPipeline (:247-264) without GUI: sequential schedule (front-end then the lag-2 back-end job)
    void run();
    // same results, the reference's two threads (front-end / back-end) with a job queue
    void run_threaded();
    std::function<void(int)> on_frame_added;            // optional hook (e.g. bench progress)
};

// ---- two-view geometry of the triangulator (vo_fivepoint.cpp), exposed for known-answer tests -------------------------------
// EMEstimatorCallback::runKernel: five NORMALISED correspondences -> up to 10 essential matrices (row-major, unit Frobenius norm)
int five_point_essentials(const double* q1, const double* q2, double* E_out);
void five_point_sample_stream(int n, int count, int* out5);   // the RANSAC's index stream (getSubset on cv::RNG((uint64)-1))
int five_point_update_num_iters(double p, double ep, int model_points, int max_iters);   // cv::RANSACUpdateNumIters
// cv::findEssentialMat(points1, points2, K, RANSAC, prob, threshold, mask) on pixel coordinates; samples_drawn counts RANSAC
// iterations; pool/pool_width: helper threads (results do not depend on them)
bool find_essential_mat(const double* p1, const double* p2, int n, const double* K, double prob, double threshold, double* E,
                        std::vector<uint8_t>& mask, int* samples_drawn, SpinPool* pool, int pool_width, FivePointTri* hook = nullptr);
// cv::recoverPose(E, points1, points2, K, R, t, HUGE_VAL, mask (in/out), triangulatedPoints): returns the number of good points
int recover_pose(FivePointTri* self, const double* E, const double* p1, const double* p2, int n, const double* K, double* R_out,
                 double* t_out, std::vector<uint8_t>& mask, std::vector<double>& tri4, bool ahead = false);
std::shared_ptr<SpinPool> make_spin_pool(int workers);

// cv::Rodrigues both ways (host copy for the adapters)
void rodrigues_v2m(const double r[3], double R[9]);
void rodrigues_m2v(const double R[9], double r[3]);

}  // namespace vo
