/* reference scenario: shared_lib_tests/test_sssp.c:12-73, ctest regex "Node ID.*1.*: Label.*39.*: Predecessor.*0"
 * (CMakeLists.txt:227-229) */
#include <stdio.h>
#include <string.h>
#include <gunrock/gunrock.h>

int main(void)
{
    struct GunrockDataType data_type = {VTXID_INT, SIZET_INT, VALUE_UINT};
    struct GunrockConfig config;
    memset(&config, 0, sizeof(config));
    config.device = 0;
    config.mark_pred = true;
    config.queue_size = 1.0f;
    config.delta_factor = 1;
    config.src_mode = manually;
    config.src_node = 0;
    int row_offsets[8] = {0, 3, 6, 9, 11, 14, 15, 15};
    int col_indices[15] = {1, 2, 3, 0, 2, 4, 3, 4, 5, 5, 6, 2, 5, 6, 6};
    unsigned int edge_values[15] = {39, 6, 41, 51, 63, 17, 10, 44, 41, 13, 58, 43, 50, 59, 35};
    struct GunrockGraph in, out;
    memset(&in, 0, sizeof(in));
    memset(&out, 0, sizeof(out));
    in.num_nodes = 7;
    in.num_edges = 15;
    in.row_offsets = row_offsets;
    in.col_indices = col_indices;
    in.edge_values = edge_values;
    int predecessor[7];
    gunrock_sssp_func(&out, predecessor, &in, config, data_type);
    unsigned int *label = (unsigned int *)out.node_values;
    printf("Demo Outputs:\n");
    for (int i = 0; i < 7; ++i) printf("Node ID [%d] : Label [%u] : Predecessor [%d]\n", i, label[i], predecessor[i]);
    int ok = label[1] == 39 && predecessor[1] == 0 && label[6] == 64;
    free(label);
    return ok ? 0 : 1;
}
