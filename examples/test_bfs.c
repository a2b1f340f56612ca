/* Known-answer driver for the BFS entry point, same scenario as the reference's shared_lib_tests/test_bfs.c:12-68
 * (7-vertex / 15-edge CSR, ctest regex "Node_ID.*2.*: Label.*1", CMakeLists.txt:215-217).  Build:
 *   gcc -std=c99 -Iinclude examples/test_bfs.c -Lgunrockinst_amd/lib -lgunrock -Wl,-rpath,$PWD/gunrockinst_amd/lib -o test_bfs */
#include <stdio.h>
#include <string.h>
#include <gunrock/gunrock.h>

int main(void)
{
    struct GunrockDataType data_type = {VTXID_INT, SIZET_INT, VALUE_INT};
    struct GunrockConfig config;
    memset(&config, 0, sizeof(config));
    config.device = 0;
    config.src_mode = manually;
    config.src_node = 0;
    config.mark_pred = false;
    config.idempotence = false;
    config.queue_size = 1.0f;

    int row_offsets[8] = {0, 3, 6, 9, 11, 14, 15, 15};
    int col_indices[15] = {1, 2, 3, 0, 2, 4, 3, 4, 5, 5, 6, 2, 5, 6, 6};
    struct GunrockGraph in, out;
    memset(&in, 0, sizeof(in));
    memset(&out, 0, sizeof(out));
    in.num_nodes = 7;
    in.num_edges = 15;
    in.row_offsets = row_offsets;
    in.col_indices = col_indices;

    gunrock_bfs_func(&out, &in, config, data_type);

    int *labels = (int *)out.node_values;
    printf("Demo Outputs:\n");
    for (int i = 0; i < 7; ++i) printf("Node_ID [%d] : Label [%d]\n", i, labels[i]);
    int ok = labels[0] == 0 && labels[2] == 1 && labels[6] == 2;
    free(labels); /* the caller owns node_values */
    return ok ? 0 : 1;
}
