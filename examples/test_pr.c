/* examples/test_pr.c -- C caller of gunrock_pr_func, used the way the reference's shared_lib_tests/test_pr.c uses it:
 * the fixed 7-vertex graph, delta 0.85, error 0.01, 20 iterations, teleport to vertex 0, the 10 best vertices.
 * The reference's ctest expects "Node ID.*2.*: Page Rank.*0.402378" (CMakeLists.txt:231-233); the code in its tree gives
 * vertex 2 the top rank with 0.3576 (tests/test_oracle.py keeps the evidence that the regex is stale), so the check
 * here is the oracle's value. */
#include <stdio.h>
#include <stdlib.h>
#include <gunrock/gunrock.h>

int main(void)
{
    struct GunrockDataType t = { VTXID_INT, SIZET_INT, VALUE_FLOAT };
    struct GunrockConfig c;
    int ro[8] = {0, 3, 6, 9, 11, 14, 15, 15};
    int ci[15] = {1, 2, 3, 0, 2, 4, 3, 4, 5, 5, 6, 2, 5, 6, 6};
    struct GunrockGraph in = {0}, out = {0};
    int top = 10, n = 7, i;
    int *ids = (int *)malloc(sizeof(int) * top);
    float *ranks = (float *)malloc(sizeof(float) * top);

    c.device = 0;
    c.delta = 0.85f;
    c.error = 0.01f;
    c.max_iter = 20;
    c.top_nodes = top;
    c.src_node = 0;
    c.src_mode = manually;
    in.num_nodes = 7;
    in.num_edges = 15;
    in.row_offsets = ro;
    in.col_indices = ci;
    gunrock_pr_func(&out, ids, ranks, &in, c, t);
    if (top > n) top = n;
    for (i = 0; i < top; ++i) printf("Node ID [%d] : Page Rank [%f]\n", ids[i], ranks[i]);
    free(ids);
    free(ranks);
    return 0;
}
