// util/test_utils.hpp -- helpers for the command-line drivers (examples/).
//
// Roles of the reference's gunrock/util/test_utils.h:50-289 and test_utils.cuh:156-190,304-405: a `--key[=value]`
// argument map, a CPU timer, a HIP-event timer and CompareResults with the same console format ("CORRECT" /
// "INCORRECT: [i]: a != b").  The CPU timer returns whole elapsed milliseconds (the reference's
// CLOCK_PROCESS_CPUTIME_ID variant drops the seconds, test_utils.h:239-249).
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace gunrock {
namespace util {

class CommandLineArgs {
    std::map<std::string, std::string> pairs;
    std::vector<std::string> positional;

   public:
    CommandLineArgs(int argc, char **argv)
    {
        for (int i = 1; i < argc; ++i) {
            std::string arg = argv[i];
            if (arg.size() < 2 || arg[0] != '-' || arg[1] != '-') {
                positional.push_back(arg);
                continue;
            }
            const size_t eq = arg.find('=');
            if (eq == std::string::npos) pairs[arg.substr(2)] = "";
            else pairs[arg.substr(2, eq - 2)] = arg.substr(eq + 1);
        }
    }
    bool CheckCmdLineFlag(const char *name) const { return pairs.find(name) != pairs.end(); }
    template <typename T>
    void GetCmdLineArgument(const char *name, T &val) const
    {
        auto it = pairs.find(name);
        if (it == pairs.end() || it->second.empty()) return;
        std::istringstream ss(it->second);
        ss >> val;
    }
    int ParsedArgc() const { return static_cast<int>(positional.size()); }
    const std::string &Positional(int i) const { return positional[i]; }
};

struct CpuTimer {
    std::chrono::steady_clock::time_point start, stop;
    void Start() { start = std::chrono::steady_clock::now(); }
    void Stop() { stop = std::chrono::steady_clock::now(); }
    double ElapsedMillis() const { return std::chrono::duration<double, std::milli>(stop - start).count(); }
};

struct GpuTimer {
    hipEvent_t start, stop;
    GpuTimer() { hipEventCreate(&start); hipEventCreate(&stop); }
    ~GpuTimer() { hipEventDestroy(start); hipEventDestroy(stop); }
    void Start(hipStream_t s = 0) { hipEventRecord(start, s); }
    void Stop(hipStream_t s = 0) { hipEventRecord(stop, s); }
    float ElapsedMillis()
    {
        float ms = 0;
        hipEventSynchronize(stop);
        hipEventElapsedTime(&ms, start, stop);
        return ms;
    }
};

// exact comparison; prints the first mismatch with its neighbourhood like test_utils.cuh:304-339
template <typename T, typename SizeT>
int CompareResults(const T *computed, const T *reference, SizeT len, bool verbose = true)
{
    int flag = 0;
    for (SizeT i = 0; i < len; ++i) {
        if (computed[i] == reference[i]) continue;
        if (flag == 0) {
            std::printf("\nINCORRECT: [%lu]: %lld != %lld", (unsigned long)i, (long long)computed[i], (long long)reference[i]);
            if (verbose) {
                const SizeT lo = i >= 5 ? i - 5 : 0, hi = i + 5 < len ? i + 5 : len;
                std::printf("\nresult[...");
                for (SizeT j = lo; j < hi; ++j) std::printf("%lld, ", (long long)computed[j]);
                std::printf("...]\nreference[...");
                for (SizeT j = lo; j < hi; ++j) std::printf("%lld, ", (long long)reference[j]);
                std::printf("...]");
            }
        }
        ++flag;
    }
    std::printf("\n");
    if (flag == 0) std::printf("CORRECT");
    return flag;
}

// Floating-point form, with the reference's tolerance (gunrock/util/test_utils.cuh:360-405): a value below 0.01 in
// magnitude may differ by 0.05 absolutely, any other by 5 % relatively.
template <typename SizeT>
int CompareResults(const float *computed, const float *reference, SizeT len, bool verbose = true)
{
    const float threshold = 0.05f;
    int flag = 0;
    for (SizeT i = 0; i < len; ++i) {
        bool is_right = true;
        const float ref = reference[i], got = computed[i];
        const float mag = ref < 0 ? -ref : ref;
        const float diff = got > ref ? got - ref : ref - got;
        if (mag < 0.01f) is_right = diff <= threshold;
        else is_right = diff <= threshold * mag;
        if (is_right) continue;
        if (flag == 0) {
            std::printf("\nINCORRECT: [%lu]: %f != %f", (unsigned long)i, got, ref);
            if (verbose) {
                const SizeT lo = i >= 5 ? i - 5 : 0, hi = i + 5 < len ? i + 5 : len;
                std::printf("\nresult[...");
                for (SizeT j = lo; j < hi; ++j) std::printf("%f, ", computed[j]);
                std::printf("...]\nreference[...");
                for (SizeT j = lo; j < hi; ++j) std::printf("%f, ", reference[j]);
                std::printf("...]");
            }
        }
        ++flag;
    }
    std::printf("\n");
    if (flag == 0) std::printf("CORRECT");
    return flag;
}

}  // namespace util
}  // namespace gunrock
