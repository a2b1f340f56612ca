/* reference scenario: shared_lib_tests/test_bc.c:12-91 (undirected 7-vertex graph, src_node = -1: every source),
 * ctest regex "Node_ID.*0.*: BC.*0.500000" (CMakeLists.txt:219-221) */
#include <stdio.h>
#include <string.h>
#include <gunrock/gunrock.h>

int main(void)
{
    struct GunrockDataType data_type = {VTXID_INT, SIZET_INT, VALUE_FLOAT};
    struct GunrockConfig config;
    memset(&config, 0, sizeof(config));
    config.device = 0;
    config.src_node = -1;
    config.queue_size = 1.0f;
    config.src_mode = manually;
    int row_offsets[8] = {0, 3, 6, 11, 15, 19, 23, 26};
    int col_indices[26] = {1, 2, 3, 0, 2, 4, 0, 1, 3, 4, 5, 0, 2, 5, 6, 1, 2, 5, 6, 2, 3, 4, 6, 3, 4, 5};
    struct GunrockGraph in, out;
    memset(&in, 0, sizeof(in));
    memset(&out, 0, sizeof(out));
    in.num_nodes = 7;
    in.num_edges = 26;
    in.row_offsets = row_offsets;
    in.col_indices = col_indices;
    gunrock_bc_func(&out, &in, config, data_type);
    float *bc = (float *)out.node_values;
    float *ebc = (float *)out.edge_values;
    if (!bc || !ebc) return 1;
    printf("Demo Outputs:\n");
    for (int i = 0; i < 7; ++i) printf("Node_ID [%d] : BC[%f]\n", i, bc[i]);
    printf("\n");
    for (int i = 0; i < 26; ++i) printf("Edge_ID [%d] : EBC[%f]\n", i, ebc[i]);
    const int ok = bc[0] > 0.4999f && bc[0] < 0.5001f;
    free(bc);
    free(ebc);
    return ok ? 0 : 1;
}
