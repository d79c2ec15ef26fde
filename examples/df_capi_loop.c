/* The reference's C API frame loop (libDF/src/capi.rs; the shape of its LADSPA / OBS-style hosts) against libdfx.so.
 *
 *   gcc -std=c99 -I include examples/df_capi_loop.c -L deepfilternet_amd/csrc -ldfx -Wl,-rpath,$PWD/deepfilternet_amd/csrc -o df_loop
 *   ./df_loop model.dfx < noisy_48k_mono_f32.raw > enhanced_f32.raw
 *
 * model.dfx: python -c "import deepfilternet_amd as d; d.export_dfx('model.dfx', '<model dir>')"
 */
#include <stdio.h>
#include <stdlib.h>

#include "df_capi.h"
#include "dfx.h"

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s model.dfx < in.f32 > out.f32   (libdfx %d, %d HIP device(s))\n", argv[0], dfx_version(), dfx_device_count());
        return 2;
    }
    DFState *st = df_create(argv[1], 100.0f, "info");
    if (!st) {
        fprintf(stderr, "df_create failed: %s\n", dfx_last_error());
        return 1;
    }
    for (char *m; (m = df_next_log_msg(st)) != NULL; df_free_log_msg(m)) fprintf(stderr, "%s\n", m);
    const size_t hop = df_get_frame_length(st);
    float *in = (float *)malloc(hop * sizeof(float)), *out = (float *)malloc(hop * sizeof(float));
    size_t frames = 0;
    while (fread(in, sizeof(float), hop, stdin) == hop) {
        const float lsnr = df_process_frame(st, in, out);
        fwrite(out, sizeof(float), hop, stdout);
        if (++frames % 100 == 0) fprintf(stderr, "frame %zu: lsnr %.1f dB\n", frames, lsnr);
    }
    free(in);
    free(out);
    df_free(st);
    return 0;
}
