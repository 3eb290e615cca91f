// api_farm.cpp -- the in-process multi-GPU farm behind the C ABI (include/avifhip.h: avifhipSetDeviceSet; SURVEY.md 8e "one host thread + one
// HIP stream + a pinned double-buffer per GPU").  libavif fans a large conversion out over row bands on host threads
// (src/reformat.c:1695-1747); this is the same fan-out across devices: a host-resident image is cut into contiguous row shares, every
// share goes to a persistent worker thread that owns a pooled context (streams, events, device scratch, download helper: api.cpp) on ITS
// device and runs the ordinary banded host-resident path on its rows.  What a share needs from beyond its rows -- the chroma filter's one
// sample row above and below -- is uploaded to that device with the share, never exchanged between devices: there is no collective and no
// peer access.  Every device moves its bytes over its own host link, which is what a host-resident call is bound by.
// The same set may name a device more than once ("0,0": two workers, two contexts, one GPU) -- how the path is exercised on a one-GPU box.
#include "api_internal.h"

#include <memory>

#include <unistd.h>

using namespace avifhip;
using namespace avifhip::api;

namespace avifhip {
namespace api {

namespace {

struct Outcome
{
    avifResult result = AVIF_RESULT_OK;
    int device = -1;
    uint64_t launches = 0, bytesUp = 0, bytesDown = 0;
    char error[sizeof(Context::lastError)] = { 0 };
    char kernel[sizeof(Context::lastKernelText)] = { 0 };
};

// one farmed call: its tasks report here, the caller waits for the last of them
struct Batch
{
    std::mutex mutex;
    std::condition_variable done;
    uint32_t pending = 0;
    std::vector<Outcome> outcomes;
};

struct Task
{
    avifResult (*job)(void * arg, uint32_t worker, FarmShare share);
    void * arg;
    uint32_t index;
    FarmShare share;
    Batch * batch;
};

class Worker
{
public:
    explicit Worker(int device) : device_(device), thread_([this] { run(); }) {}
    ~Worker()
    {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            stop_ = true;
        }
        wake_.notify_all();
        if (thread_.joinable())
            thread_.join();
    }
    void post(const Task & task)
    {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            queue_.push_back(task);
        }
        wake_.notify_one();
    }

private:
    void run()
    {
        // the thread's context lives on this worker's device from here on (ensureContext switches back should anything change it)
        const avifResult ready = avifhipSetDevice(device_);
        char readyError[sizeof(Context::lastError)];
        snprintf(readyError, sizeof(readyError), "%s", tls.lastError);
        for (;;) {
            Task task;
            {
                std::unique_lock<std::mutex> lock(mutex_);
                wake_.wait(lock, [this] { return stop_ || !queue_.empty(); });
                if (queue_.empty())
                    return; // stop requested and nothing left
                task = queue_.front();
                queue_.pop_front();
            }
            Outcome o;
            o.device = device_;
            if (ready != AVIF_RESULT_OK) {
                o.result = ready;
                snprintf(o.error, sizeof(o.error), "farm worker on device %d: %s", device_, readyError);
            } else {
                Context & c = tls;
                const uint64_t launches0 = c.launches;
                c.bytesUp = c.bytesDown = 0;
                c.lastError[0] = 0;
                o.result = task.job(task.arg, task.index, task.share);
                o.launches = c.launches - launches0;
                o.bytesUp = c.bytesUp, o.bytesDown = c.bytesDown;
                snprintf(o.error, sizeof(o.error), "%s", c.lastError);
                snprintf(o.kernel, sizeof(o.kernel), "%s", c.lastKernel ? c.lastKernel : "");
            }
            {
                std::lock_guard<std::mutex> lock(task.batch->mutex);
                task.batch->outcomes[task.index] = o;
                if (--task.batch->pending == 0)
                    task.batch->done.notify_all();
            }
        }
    }
    int device_;
    std::mutex mutex_;
    std::condition_variable wake_;
    std::deque<Task> queue_;
    bool stop_ = false;
    std::thread thread_; // (last: started when everything above exists)
};

struct Farm
{
    std::vector<int> devices;
    std::vector<std::unique_ptr<Worker>> workers;
    int pid = 0;
};

struct FarmState
{
    std::mutex mutex;
    std::vector<int> deviceSet; // what avifhipSetDeviceSet / AVIFHIP_DEVICES asked for
    bool decided = false;       // the environment has been looked at, or the API was called
    Farm * farm = nullptr;      // workers of `deviceSet`, started at the first farmed call
};
// never destroyed, like the context pool: worker threads may still be parked on their condition variables when the process exits
FarmState & farmState()
{
    static FarmState * state = new FarmState;
    return *state;
}

// AVIFHIP_DEVICES = "all" | "<n>,<n>,..." (non-negative device indices; anything else is ignored: no farm)
std::vector<int> deviceSetFromEnvironment()
{
    std::vector<int> set;
    const char * e = getenv("AVIFHIP_DEVICES");
    if (!e || !*e)
        return set;
    const int n = avifhipDeviceCount();
    if (!strcmp(e, "all")) {
        for (int d = 0; d < n; ++d)
            set.push_back(d);
        return set;
    }
    for (const char * p = e;;) {
        if (*p < '0' || *p > '9')
            return std::vector<int>();
        long v = 0;
        while (*p >= '0' && *p <= '9' && v < 4096)
            v = v * 10 + (*p++ - '0');
        if (v >= 4096)
            return std::vector<int>();
        if (v >= n) {
            // like avifhipSetDeviceSet, which refuses such a set: a worker on a device that does not exist would fail every farmed call
            //  -- the set is ignored, calls run on the calling thread's device, and the user is told once
            fprintf(stderr, "avifhip: AVIFHIP_DEVICES=%s names device %ld, but %d HIP device(s) are visible: the device set is ignored\n", e, v, n);
            return std::vector<int>();
        }
        set.push_back((int)v);
        if (!*p)
            return set;
        if (*p++ != ',')
            return std::vector<int>();
    }
}

void setMinShareFromEnvironment();

// (state.mutex held)
void decide(FarmState & state)
{
    if (!state.decided) {
        state.deviceSet = deviceSetFromEnvironment();
        setMinShareFromEnvironment();
        state.decided = true;
    }
}

// (state.mutex held) stops the workers of a set that is no longer current; their queued tasks are finished first
void retire(FarmState & state)
{
    if (!state.farm)
        return;
    if (state.farm->pid == (int)getpid())
        delete state.farm; // joins the workers
    // (a fork()ed child has no such threads: the parent's farm object is left alone)
    state.farm = nullptr;
}

} // namespace

// smallest share worth a device of its own, in pixels (avifhipSetFarmMinSharePixels; default 2^21)
static std::atomic<uint64_t> gFarmMinShare { (uint64_t)1 << 21 };

namespace {
// AVIFHIP_FARM_MIN_PIXELS=<n> beside AVIFHIP_DEVICES: the same knob for processes that cannot call the API (tests drive the reference's own
// programs over seam A / seam B with small fixtures through the farm)
void setMinShareFromEnvironment()
{
    const char * e = getenv("AVIFHIP_FARM_MIN_PIXELS");
    if (!e || !*e)
        return;
    uint64_t v = 0;
    for (const char * p = e; *p; ++p) {
        if (*p < '0' || *p > '9' || v > ((uint64_t)1 << 40))
            return;
        v = v * 10 + (uint64_t)(*p - '0');
    }
    if (v)
        gFarmMinShare.store(v, std::memory_order_relaxed);
}
} // namespace

std::vector<FarmShare> planFarmRows(uint32_t width, uint32_t height, uint32_t workers)
{
    std::vector<FarmShare> shares;
    const uint64_t pixels = (uint64_t)width * height;
    const uint64_t byPixels = pixels / gFarmMinShare.load(std::memory_order_relaxed); // ~2 megapixels per share at least
    uint32_t usable = workers < 1 ? 1 : workers;
    if ((uint64_t)usable > byPixels)
        usable = byPixels < 1 ? 1u : (uint32_t)byPixels;
    uint32_t rows = (height + usable - 1) / usable;
    rows = (rows + 31u) & ~31u;
    if (usable <= 1 || rows >= height) {
        shares.push_back({ 0, height });
        return shares;
    }
    for (uint32_t y = 0; y < height; y += rows)
        shares.push_back({ y, (height - y > rows) ? y + rows : height });
    return shares;
}

std::vector<FarmShare> planFarmJobs(uint32_t count, uint32_t workers)
{
    std::vector<FarmShare> shares;
    if (workers < 1)
        workers = 1;
    const uint32_t base = count / workers, extra = count % workers;
    uint32_t first = 0;
    for (uint32_t k = 0; k < workers && first < count; ++k) {
        const uint32_t n = base + (k < extra ? 1u : 0u);
        if (n)
            shares.push_back({ first, first + n });
        first += n;
    }
    return shares;
}

uint64_t farmMinSharePixels()
{
    const uint64_t v = gFarmMinShare.load(std::memory_order_relaxed);
    return v ? v : 1;
}

uint32_t farmWorkers()
{
    FarmState & state = farmState();
    std::lock_guard<std::mutex> lock(state.mutex);
    decide(state);
    return (uint32_t)state.deviceSet.size();
}

avifResult farmRun(const std::vector<FarmShare> & shares, avifResult (*job)(void * arg, uint32_t worker, FarmShare share), void * arg)
{
    Batch batch;
    bool runHere = false;
    const uint32_t n = (uint32_t)shares.size();
    batch.pending = n;
    batch.outcomes.resize(n);
    {
        FarmState & state = farmState();
        std::lock_guard<std::mutex> lock(state.mutex);
        decide(state);
        if (state.farm && (state.farm->pid != (int)getpid() || state.farm->devices != state.deviceSet))
            retire(state);
        if (n > state.deviceSet.size())
            runHere = true; // the set shrank between the caller's plan and now (avifhipSetDeviceSet from another thread): see below
        if (!runHere && !state.farm) {
            state.farm = new Farm;
            state.farm->devices = state.deviceSet;
            state.farm->pid = (int)getpid();
            for (int d : state.deviceSet)
                state.farm->workers.emplace_back(new Worker(d));
        }
        for (uint32_t k = 0; !runHere && k < n; ++k)
            state.farm->workers[k]->post({ job, arg, k, shares[k], &batch });
    }
    if (runHere) {
        // the shares are still a valid cut of the work: the calling thread takes them one after the other on its own device
        tls.farmReports.clear();
        for (uint32_t k = 0; k < n; ++k) {
            const avifResult r = job(arg, k, shares[k]);
            if (r != AVIF_RESULT_OK)
                return r;
        }
        return AVIF_RESULT_OK;
    }
    {
        std::unique_lock<std::mutex> lock(batch.mutex);
        batch.done.wait(lock, [&batch] { return batch.pending == 0; });
    }
    // AVIFHIP_FARM_TRACE: one line on stderr per farmed call (which devices took which shares) -- how a test of an unmodified application sees
    // that the device set from its environment was used
    static const bool trace = getenv("AVIFHIP_FARM_TRACE") != nullptr;
    if (trace) {
        fprintf(stderr, "avifhip farm: %u shares", n);
        for (uint32_t k = 0; k < n; ++k)
            fprintf(stderr, " [%u,%u)@%d", shares[k].begin, shares[k].end, batch.outcomes[k].device);
        fprintf(stderr, "\n");
    }
    Context & c = tls;
    c.farmReports.clear();
    c.bytesUp = c.bytesDown = 0;
    avifResult result = AVIF_RESULT_OK;
    for (uint32_t k = 0; k < n; ++k) {
        const Outcome & o = batch.outcomes[k];
        c.farmReports.push_back({ o.device, shares[k].begin, shares[k].end, o.bytesUp, o.bytesDown });
        c.launches += o.launches;
        c.bytesUp += o.bytesUp, c.bytesDown += o.bytesDown;
        if (o.result != AVIF_RESULT_OK && result == AVIF_RESULT_OK) {
            result = o.result;
            snprintf(c.lastError, sizeof(c.lastError), "%s", o.error);
        }
    }
    if (n) {
        snprintf(c.lastKernelText, sizeof(c.lastKernelText), "%s", batch.outcomes[0].kernel);
        c.lastKernel = c.lastKernelText;
    }
    return result;
}

} // namespace api
} // namespace avifhip

extern "C" avifResult avifhipSetDeviceSet(const int * devices, uint32_t count)
{
    if (count && !devices)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const int visible = avifhipDeviceCount();
    for (uint32_t k = 0; k < count; ++k) {
        if (devices[k] < 0 || (visible > 0 && devices[k] >= visible)) {
            setError("avifhipSetDeviceSet: device %d is not one of the %d visible HIP device(s)", devices[k], visible);
            return AVIF_RESULT_INVALID_ARGUMENT;
        }
    }
    FarmState & state = farmState();
    std::lock_guard<std::mutex> lock(state.mutex);
    state.deviceSet.assign(devices, devices + count);
    state.decided = true;
    if (state.farm && state.farm->devices != state.deviceSet)
        retire(state);
    return AVIF_RESULT_OK;
}

extern "C" void avifhipSetFarmMinSharePixels(uint64_t pixels)
{
    gFarmMinShare.store(pixels ? pixels : (uint64_t)1 << 21, std::memory_order_relaxed);
}

extern "C" uint32_t avifhipGetDeviceSet(int * devices, uint32_t capacity)
{
    FarmState & state = farmState();
    std::lock_guard<std::mutex> lock(state.mutex);
    decide(state);
    for (uint32_t k = 0; k < capacity && k < state.deviceSet.size(); ++k)
        devices[k] = state.deviceSet[k];
    return (uint32_t)state.deviceSet.size();
}

extern "C" avifResult avifhipPlanFarmRows(uint32_t width, uint32_t height, uint32_t workers, avifCropRect * bands, uint32_t capacity, uint32_t * count)
{
    if (!count || !width || !height)
        return AVIF_RESULT_INVALID_ARGUMENT;
    const std::vector<FarmShare> shares = planFarmRows(width, height, workers);
    *count = (uint32_t)shares.size();
    if (bands) {
        if (capacity < shares.size())
            return AVIF_RESULT_INVALID_ARGUMENT;
        for (size_t k = 0; k < shares.size(); ++k)
            bands[k] = { 0, shares[k].begin, width, shares[k].end - shares[k].begin };
    }
    return AVIF_RESULT_OK;
}

extern "C" uint32_t avifhipLastFarmWorkers(void)
{
    return (uint32_t)tls.farmReports.size();
}

extern "C" avifResult avifhipLastFarmTransferBytes(uint32_t worker, int * device, uint32_t * begin, uint32_t * end, uint64_t * bytesUp, uint64_t * bytesDown)
{
    const Context & c = tls;
    if (worker >= c.farmReports.size())
        return AVIF_RESULT_INVALID_ARGUMENT;
    const Context::FarmReport & r = c.farmReports[worker];
    if (device)
        *device = r.device;
    if (begin)
        *begin = r.rowBegin;
    if (end)
        *end = r.rowEnd;
    if (bytesUp)
        *bytesUp = r.bytesUp;
    if (bytesDown)
        *bytesDown = r.bytesDown;
    return AVIF_RESULT_OK;
}
