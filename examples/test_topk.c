/* examples/test_topk.c -- C caller of gunrock_topk_func with the inputs of the reference's shared_lib_tests/test_topk.c
 * (the 7-vertex graph as CSR and CSC, the 3 vertices of largest in + out degree).
 * Known answer (reference CMakeLists.txt:235-237): "Node ID.*2.*: in_degrees.*3.*: out_degrees.*3". */
#include <stdio.h>
#include <gunrock/gunrock.h>

int main(void)
{
    struct GunrockDataType t = { VTXID_INT, SIZET_INT, VALUE_INT };
    struct GunrockConfig c;
    int ro[8] = {0, 3, 6, 9, 11, 14, 15, 15};
    int ci[15] = {1, 2, 3, 0, 2, 4, 3, 4, 5, 5, 6, 2, 5, 6, 6};
    int co[8] = {0, 1, 2, 5, 7, 9, 12, 15};
    int ri[15] = {1, 0, 0, 1, 4, 0, 2, 1, 2, 2, 3, 4, 3, 4, 5};
    struct GunrockGraph in = {0}, out = {0};
    int ids[3], ind[3], outd[3], i;

    c.device = 0;
    c.top_nodes = 3;
    in.num_nodes = 7;
    in.num_edges = 15;
    in.row_offsets = ro;
    in.col_indices = ci;
    in.col_offsets = co;
    in.row_indices = ri;
    gunrock_topk_func(&out, ids, ind, outd, &in, c, t);
    for (i = 0; i < 3; ++i) printf("Node ID [%d] : in_degrees [%d] : out_degrees [%d]\n", ids[i], ind[i], outd[i]);
    return 0;
}
