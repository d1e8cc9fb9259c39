// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h). Declarations of the speed-oriented twins in orc_fast.cpp.
#pragma once
#include "orc_api.h"
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace orc {

// Persistent worker threads. parallel_for splits [0, n) into contiguous chunks that the workers and the caller take by ticket;
// tickets carry the batch generation, so a worker that wakes up late can never touch the counters or the closure of a later batch.
class Pool {
public:
    explicit Pool(int workers);
    ~Pool();
    int threads() const { return (int)th_.size() + 1; }
    // fn(lo, hi) over at most max_chunks (0 = 4 per thread) chunks; returns when every chunk is done
    void parallel_for(int n, int max_chunks, const std::function<void(int, int)>& fn);
private:
    void worker(int id);
    void drain(unsigned gen);
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    unsigned gen_ = 0;
    bool stop_ = false;
    std::atomic<unsigned long long> ticket_{0};   // (generation << 32) | next chunk
    std::atomic<int> done_{0};
    int n_ = 0, nchunk_ = 0;
    const std::function<void(int, int)>* fn_ = nullptr;
};

void pyr_down_fast(const Image8& src, Image8& dst);
void lk_track_fast(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_xy, int n, const LKParams& P,
                   float* out_xy, uint8_t* out_status, float* out_err, Pool* pool);
// BA: residual/Jacobian evaluation on `pool` split into `threads` ranges (results combined in observation order); nullptr = serial
void ba_set_pool(Pool* pool, int threads);

}  // namespace orc
