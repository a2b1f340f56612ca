// lib/stubs.hip -- entry points of gunrock.h that are outside this build's scope (SURVEY.md section 8:
// PageRank and TopK are other primitives).  Exported so programs written against the reference
// header keep linking; they fail loudly instead of computing anything.
#include <gunrock/gunrock.h>

#include <cstdio>

extern "C" {

void gunrock_pr_func(struct GunrockGraph *, void *, void *, const struct GunrockGraph *, struct GunrockConfig,
                     struct GunrockDataType)
{
    std::fprintf(stderr, "[gunrock-mi355x] gunrock_pr_func is not built in this library (BFS/CC/SSSP/BC only).\n");
}

void gunrock_topk_func(struct GunrockGraph *, void *, void *, void *, const struct GunrockGraph *, struct GunrockConfig,
                       struct GunrockDataType)
{
    std::fprintf(stderr, "[gunrock-mi355x] gunrock_topk_func is not built in this library (BFS/CC/SSSP/BC only).\n");
}

}  // extern "C"
