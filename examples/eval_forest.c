/* A plain-C consumer of librdf_hip.so: what the drop-in boundary looks like without Python.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/eval_forest.c \
 *       -L 3d-beats_amd/csrc -l:librdf_hip.so -L /opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/3d-beats_amd/csrc -Wl,-rpath,/opt/rocm/lib -o eval_forest_example && ./eval_forest_example
 *
 * One 8x6 depth frame, one tree of depth 1 (the root, both sides leaves).  The root compares the pixel with itself
 * (u = v = 0), so f = 0 < thresh = 1 sends every valid pixel left, whose PDF (0.2, 0.8) makes label 1; pixels with
 * depth 0 or 65535 keep the caller's pre-fill (tree_eval.cu:88-89). */
#include <stdio.h>
#include <stdint.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "rdf_hip.h"

#define W 8
#define H 6

int main(void)
{
    uint16_t depth[H * W], labels[H * W];
    float forest[1 * 1 * (7 + 2 * 2)] = {0.f, 0.f, 0.f, 0.f, /* thresh */ 1.f, /* l_next, r_next: leaves */ 0.f, 0.f,
                                         /* left pdf */ 0.2f, 0.8f, /* right pdf */ 0.9f, 0.1f};
    void *d_depth = NULL, *d_labels = NULL, *d_forest = NULL;
    int i, rc, bad = 0;

    if (rdf_abi_version() < 1) return 2;
    for (i = 0; i < H * W; ++i) depth[i] = (uint16_t)(3000 + i);
    depth[5] = 0;
    depth[17] = 65535;
    memset(labels, 0xFF, sizeof labels);               /* the caller's pre-fill: 65535 */

    if (hipMalloc(&d_depth, sizeof depth) != hipSuccess || hipMalloc(&d_labels, sizeof labels) != hipSuccess ||
        hipMalloc(&d_forest, sizeof forest) != hipSuccess) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    hipMemcpy(d_depth, depth, sizeof depth, hipMemcpyHostToDevice);
    hipMemcpy(d_labels, labels, sizeof labels, hipMemcpyHostToDevice);
    hipMemcpy(d_forest, forest, sizeof forest, hipMemcpyHostToDevice);

    rc = rdf_eval_forest((const uint16_t *)d_depth, 1, W, H, (const float *)d_forest, 1, 1, 2, NULL, -1,
                         (uint16_t *)d_labels, 1, 1.0f, NULL);
    if (rc != RDF_OK) {
        fprintf(stderr, "rdf_eval_forest: %s\n", rdf_error_string(rc));
        return 1;
    }
    hipDeviceSynchronize();
    hipMemcpy(labels, d_labels, sizeof labels, hipMemcpyDeviceToHost);
    for (i = 0; i < H * W; ++i) {
        const unsigned want = (i == 5 || i == 17) ? 65535u : 1u;
        if (labels[i] != want) ++bad;
    }
    /* argument errors come back as negative codes, not as faults */
    if (rdf_eval_forest(NULL, 1, W, H, (const float *)d_forest, 1, 1, 2, NULL, -1, (uint16_t *)d_labels, 1, 1.0f, NULL) !=
        RDF_ERR_NULL_PTR)
        ++bad;
    hipFree(d_depth); hipFree(d_labels); hipFree(d_forest);
    printf(bad ? "FAIL: %d mismatches\n" : "PASS\n", bad);
    return bad ? 1 : 0;
}
