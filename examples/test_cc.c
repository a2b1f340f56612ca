/* reference scenario: shared_lib_tests/test_cc.c:12-63, ctest regex "Node_ID.*1.*: Component_ID.*0" (CMakeLists.txt:223-225) */
#include <stdio.h>
#include <string.h>
#include <gunrock/gunrock.h>

int main(void)
{
    struct GunrockDataType data_type = {VTXID_INT, SIZET_INT, VALUE_INT};
    struct GunrockConfig config;
    memset(&config, 0, sizeof(config));
    config.device = 0;
    int row_offsets[8] = {0, 3, 6, 9, 11, 14, 15, 15};
    int col_indices[15] = {1, 2, 3, 0, 2, 4, 3, 4, 5, 5, 6, 2, 5, 6, 6};
    struct GunrockGraph in, out;
    memset(&in, 0, sizeof(in));
    memset(&out, 0, sizeof(out));
    in.num_nodes = 7;
    in.num_edges = 15;
    in.row_offsets = row_offsets;
    in.col_indices = col_indices;
    gunrock_cc_func(&out, &in, config, data_type);
    int *ids = (int *)out.node_values;
    printf("Demo Outputs:\n");
    int ok = 1;
    for (int i = 0; i < 7; ++i) {
        printf("Node_ID [%d] : Component_ID [%d]\n", i, ids[i]);
        ok = ok && ids[i] == 0;
    }
    free(ids);
    return ok ? 0 : 1;
}
