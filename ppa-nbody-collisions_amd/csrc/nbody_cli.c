/* csrc/nbody_cli.c -- `nbody`: command-line driver with the reference binary's behaviour on the hot path
 * (main(), src/nbody.cu:373-551): reads ./nbodyConfig.txt from the current directory (:377), echoes the
 * settings, seeds the bodies (:401-416), runs totalIterations steps and prints the elapsed time (:548).
 * Image output (:512-539) is produced with --images: iteration_<k>.ppm in cfg.imagePath for every k that is a
 * multiple of save_Image_Every_Xth_Iteration and not the last iteration (the reference saves the image of
 * iteration k during iteration k+1, :513-522, so the last one is never written).  Without --images the run is
 * the pure stepping loop.  Host code in C over the C ABI. */
#include "nbody.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

static double now(void) {   /* jbutil::gettime, include/jbutil.h:98-104 */
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (double)tv.tv_sec + (double)tv.tv_usec * 1e-6;
}

static int die(const char* what, int rc) {
    fprintf(stderr, "%s: %s: %s\n", what, nbody_status_string(rc), nbody_last_error_string());
    return 1;
}

int main(int argc, char** argv) {
    const char* path = "nbodyConfig.txt";
    int precision = NBODY_F32, gpus = 1, dump = 0, images = 0;
    for (int a = 1; a < argc; ++a) {
        if (!strcmp(argv[a], "--config") && a + 1 < argc) path = argv[++a];
        else if (!strcmp(argv[a], "--fp64")) precision = NBODY_F64;
        else if (!strcmp(argv[a], "--gpus") && a + 1 < argc) gpus = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--dump")) dump = 1;
        else if (!strcmp(argv[a], "--images")) images = 1;
        else { fprintf(stderr, "usage: nbody [--config FILE] [--fp64] [--gpus N] [--dump] [--images]\n"); return 2; }
    }
    if (gpus < 1 || gpus > 64) return 2;
    double startTime = now();
    printf("Running simulation with the following settings:\n");
    nbody_config cfg;
    int rc = nbody_config_parse(path, &cfg);
    if (rc != NBODY_OK) return 1;                     /* the reference exit(1)s here */
    printf("=====================\n");
    void* block = nbody_block_alloc(cfg.particleCount, precision);
    if (!block) return die("alloc", NBODY_ERR_NOMEM);
    printf("Bodies: %d\n", cfg.particleCount);
    rc = nbody_init_bodies(&cfg, block, precision);
    if (rc != NBODY_OK) return die("init", rc);

    nbody_ctx* ctxs[64];
    for (int g = 0; g < gpus; ++g) {
        nbody_ctx_desc d;
        nbody_ctx_desc_from_config(&d, &cfg, precision);
        d.device = g; d.rank = g; d.world = gpus;
        d.flags = gpus > 1 ? NBODY_FLAG_GROUP_EXCHANGE : 0;
        rc = nbody_ctx_create(&ctxs[g], &d);
        if (rc != NBODY_OK) return die("ctx_create", rc);
        rc = nbody_upload(ctxs[g], block, cfg.particleCount);
        if (rc != NBODY_OK) return die("upload", rc);
    }
    double t0 = now();
    if (!images || cfg.save_Image_Every_Xth_Iteration <= 0) {
        rc = nbody_group_step(ctxs, gpus, cfg.totalIterations);
        if (rc != NBODY_OK) return die("step", rc);
    } else {
        unsigned char* img = (unsigned char*)malloc((size_t)cfg.imgWidth * cfg.imgHeight);
        if (!img) return die("image alloc", NBODY_ERR_NOMEM);
        int done = 0;
        while (done < cfg.totalIterations) {
            /* next iteration whose image the reference would save: k % every == 0 and k + 1 < totalIterations */
            int k = ((done + cfg.save_Image_Every_Xth_Iteration - 1) / cfg.save_Image_Every_Xth_Iteration) *
                    cfg.save_Image_Every_Xth_Iteration;
            int upto = (k + 1 < cfg.totalIterations) ? k + 1 : cfg.totalIterations;
            rc = nbody_group_step(ctxs, gpus, upto - done);
            if (rc != NBODY_OK) return die("step", rc);
            done = upto;
            if (k + 1 < cfg.totalIterations && done == k + 1) {
                char name[NBODY_IMAGE_PATH_MAX + 64];
                rc = nbody_render_image(ctxs[0], img, cfg.imgWidth, cfg.imgHeight);
                if (rc != NBODY_OK) return die("render", rc);
                snprintf(name, sizeof(name), "%s/iteration_%d.ppm", cfg.imagePath, k);      /* :518 */
                if (nbody_write_pgm(name, img, cfg.imgWidth, cfg.imgHeight) != NBODY_OK) return 1;   /* :369 */
            }
        }
        free(img);
    }
    int n = 0;
    rc = nbody_group_download(ctxs, gpus, block, &n);
    if (rc != NBODY_OK) return die("download", rc);
    double t1 = now();
    long long pairs = 0;
    for (int g = 0; g < gpus; ++g) {
        nbody_stats s;
        nbody_get_stats(ctxs[g], &s);
        pairs += s.pairs;
        nbody_ctx_destroy(ctxs[g]);
    }
    printf("Bodies left: %d\n", n);
    printf("Stepping: %.4f s, %.4e body-pair-interactions/sec\n", t1 - t0, (double)pairs / (t1 - t0));
    if (dump) fwrite(block, 1, nbody_block_bytes(n, precision), stdout);
    nbody_block_free(block);
    printf("Time taken: %.4f\n", now() - startTime);   /* src/nbody.cu:548 */
    return 0;
}
