// Per-kernel HIP-event timing (bench.py's roofline leg): events are recorded around every launch of a kernel class on the
// stream it is launched on and read back after the timed region; nothing synchronises while recording.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

namespace pmv {

enum KernelId { K_PAD0 = 0, K_PYRDOWN, K_LK, K_GFTT_CAND, K_GFTT_PICK, K_ST_RESP, K_ST_SELECT, K_PNP_HYP, K_PNP_REFIT,
                K_BA_LM, K_BA_RESID, K_TRI_DLT, K_BAM_EVAL0, K_BAM_CAMPOINT, K_BAM_GEMM, K_BAM_SOLVE, K_BAM_BACKSUB, K_BAM_FINISH, K_FIVEPOINT, K_COUNT };

inline const char* kernel_name(int id) {
    static const char* n[K_COUNT] = {"k_pad_level0", "k_pyrdown", "k_lk", "k_gftt_cand", "k_gftt_pick", "k_st_resp", "k_st_select",
                                     "k_pnp_hyp", "k_pnp_select_refit", "ba_lm_chain", "k_ba_residuals", "k_tri_dlt",
                                     "k_bam_eval0", "k_bam_campoint", "k_bam_gemm", "k_bam_solve", "k_bam_backsub", "k_bam_finish", "k_fivepoint_hyp+score"};
    return (id >= 0 && id < K_COUNT) ? n[id] : "?";
}

// "ba_lm_chain" times one whole LM solve: the chain of k_bam_* launches (or k_ba_lm in single-workgroup mode); every other
// class is exactly one kernel per launch. The k_bam_* classes time the kernels of the chain one by one; they are recorded only
// when their bit is selected explicitly (pmv_prof_select): events between the launches of a chain lengthen the chain itself.
struct Profiler {
    static constexpr int CHUNK = 8192;       // the event pool of a class grows by this many launches at a time ...
    static constexpr int CAP = 1 << 20;      // ... up to this many launches between resets (beyond it: counted in `dropped`)
    bool enabled = false;
    unsigned mask = ~0u;               // kernel classes that are recorded while enabled (bit = KernelId)
    bool chain_detail = false;         // also record the k_bam_* kernels inside a chain (set by pmv_prof_select when one is named)
    std::vector<hipEvent_t> ev[K_COUNT];
    int used[K_COUNT] = {0};
    long dropped[K_COUNT] = {0};
    hipEvent_t* next(int id) {
        if (used[id] + 2 > 2 * CAP) { dropped[id]++; return nullptr; }
        if (used[id] + 2 > (int)ev[id].size()) {
            const size_t old = ev[id].size();
            ev[id].resize(old + 2 * CHUNK);
            for (size_t i = old; i < ev[id].size(); i++) (void)hipEventCreate(&ev[id][i]);
        }
        hipEvent_t* p = &ev[id][used[id]];
        used[id] += 2;
        return p;
    }
    void destroy() {
        for (auto& v : ev) { for (auto& e : v) (void)hipEventDestroy(e); v.clear(); }
    }
};

extern thread_local Profiler* tl_prof;   // set by the C-ABI entry points for the calling thread

struct ProfScope {
    hipEvent_t* e = nullptr;
    hipStream_t s;
    ProfScope(int id, hipStream_t stream) : s(stream) {
        if (tl_prof && tl_prof->enabled && ((tl_prof->mask >> id) & 1u) && (id < K_BAM_EVAL0 || id > K_BAM_FINISH || tl_prof->chain_detail)) {
            e = tl_prof->next(id);
            if (e) (void)hipEventRecord(e[0], s);
        }
    }
    ~ProfScope() { if (e) (void)hipEventRecord(e[1], s); }
};

}  // namespace pmv
