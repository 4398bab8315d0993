/* ORACLE (test infrastructure + the timed CPU baseline; never the product path).
 *
 * POSIX fp64 restatement of the reference's CPU tile rasteriser:
 *   gso_rasterize_tile()  <- rasterize.c:27-80 (rasterize_tile) + :82-132 (thread_rasterize)
 *   gso_rasterize_frame() <- the per-tile driver loop of c_rasterize, rasterization.py:127-146
 *
 * The original cannot be compiled on Linux (#include <windows.h>, CreateThread,
 * rasterize.c:2,65-77), so this is a restatement: same blend loop, same constants (.99, 1/255,
 * 1e-4), same [x][y][3] chunk layout, same `contributor` bookkeeping (counts skipped entries and
 * the entry that triggers the break, rasterize.c:99-125).  Differences, all deliberate:
 *   - conics are double (the original declares float and is handed float64 memory: defect B7);
 *   - arguments are passed on the stack, not through file-scope globals (rasterize.c:12-23), so
 *     it is re-entrant and tiles can run on OpenMP threads (the original uses one Win32 thread per
 *     tile row, rasterize.c:50-74);
 *   - the pixel-centre offset is a parameter: +0.5 reproduces rasterize.c:103-104
 *     (dist = c - x + .5), -0.5 reproduces rasterization.py:187-191 (dist = c - (x + .5)), 0
 *     reproduces the OpenCL forward kernel setup_gpu.py:75-76.
 *   - optional `fragile` output: per pixel, 1 if some visited pair sat within a relative margin
 *     of a discontinuity (power>0, alpha<1/255, T<1e-4) -- an fp32 device path may legitimately
 *     take the other branch there; parity tests exclude or count those pixels explicitly.
 * Pinned by tests/golden/scene_s{0..3}.npz (images produced by the reference's own rasterize()).
 */
#include <math.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GSO_TILE 16

/* One tile. image_chunk[16][16][3] and last_contributors[16][16] are indexed [loc_x][loc_y]
 * exactly as rasterize.c:93,130.  Returns 0 (rasterize.c:79). */
int gso_rasterize_tile(double image_chunk[GSO_TILE][GSO_TILE][3],
                       int last_contributors[GSO_TILE][GSO_TILE],
                       double t_final_chunk[GSO_TILE][GSO_TILE],       /* may be NULL */
                       unsigned char fragile_chunk[GSO_TILE][GSO_TILE], /* may be NULL */
                       const double (*centers)[2], const double (*colors)[3],
                       const double *opacities, const double (*conics)[3],
                       const int *gaus_stack, const double background_color[3],
                       int stack_size, int tile_x, int tile_y, int result_x, int result_y,
                       double pixel_offset, double margin)
{
    const int start_x = tile_x * GSO_TILE, start_y = tile_y * GSO_TILE;
    const int columns = ((tile_x + 1) * GSO_TILE > result_x) ? (result_x - tile_x * GSO_TILE) : GSO_TILE;
    const int rows = ((tile_y + 1) * GSO_TILE > result_y) ? (result_y - tile_y * GSO_TILE) : GSO_TILE;

    for (int loc_y = 0; loc_y < rows; loc_y++) {           /* one Win32 thread per row in the original */
        const int y = loc_y + start_y;
        for (int loc_x = 0; loc_x < columns; loc_x++) {
            const int x = loc_x + start_x;
            int contributor = 0;
            double T = 1.0;
            double acc[3] = {0.0, 0.0, 0.0};
            unsigned char frag = 0;
            for (int i = 0; i < stack_size; i++) {
                contributor++;
                const int gaus = gaus_stack[i];
                const double dist_x = centers[gaus][0] - ((double)x) + pixel_offset;
                const double dist_y = centers[gaus][1] - ((double)y) + pixel_offset;
                const double qa = conics[gaus][0] * dist_x * dist_x;
                const double qc = conics[gaus][2] * dist_y * dist_y;
                const double qb = conics[gaus][1] * dist_x * dist_y;
                const double power = -.5 * (qa + qc) - qb;
                if (margin > 0.0 && fabs(power) <= margin * (fabs(.5 * qa) + fabs(.5 * qc) + fabs(qb) + 1e-30))
                    frag = 1;
                if (power > 0)
                    continue;
                const double alpha = fmin(.99, opacities[gaus] * exp(power));
                if (margin > 0.0 && fabs(alpha - 1.0 / 255) <= margin * (1.0 / 255))
                    frag = 1;
                if (alpha < (1.0 / 255))
                    continue;
                const double test_T = T * (1.0 - alpha);
                if (margin > 0.0 && fabs(test_T - .0001) <= margin * .0001)
                    frag = 1;
                if (test_T < .0001)
                    break;
                for (int ch = 0; ch < 3; ch++)
                    acc[ch] += colors[gaus][ch] * alpha * T;
                T = test_T;
            }
            for (int ch = 0; ch < 3; ch++)
                image_chunk[loc_x][loc_y][ch] = acc[ch] + background_color[ch] * T;
            last_contributors[loc_x][loc_y] = contributor;
            if (t_final_chunk) t_final_chunk[loc_x][loc_y] = T;
            if (fragile_chunk) fragile_chunk[loc_x][loc_y] = frag;
        }
    }
    return 0;
}

/* Whole frame: image[W][H][3], last_contributors[W][H], t_final[W][H] (NULL ok),
 * fragile[W][H] (NULL ok).  ranges[tiles][2] = [start,end) into vals for tile_id = tx*tiles_y+ty.
 * One tile per OpenMP task; nthreads<=0 -> OpenMP default.  Returns the thread count used. */
int gso_rasterize_frame(double *image, int *last_contributors, double *t_final,
                        unsigned char *fragile,
                        const double *centers, const double *colors, const double *opacities,
                        const double *conics, const int *vals, const int *ranges,
                        const double background_color[3], int result_x, int result_y,
                        double pixel_offset, double margin, int nthreads)
{
    const int tiles_x = (result_x + GSO_TILE - 1) / GSO_TILE;
    const int tiles_y = (result_y + GSO_TILE - 1) / GSO_TILE;
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < tiles_x * tiles_y; t++) {
        const int tx = t / tiles_y, ty = t % tiles_y;
        double chunk[GSO_TILE][GSO_TILE][3];
        int lc[GSO_TILE][GSO_TILE];
        double tf[GSO_TILE][GSO_TILE];
        unsigned char fr[GSO_TILE][GSO_TILE];
        const int s = ranges[2 * t], e = ranges[2 * t + 1];
        gso_rasterize_tile(chunk, lc, tf, fr, (const double (*)[2])centers,
                           (const double (*)[3])colors, opacities, (const double (*)[3])conics,
                           vals + s, background_color, e - s, tx, ty, result_x, result_y,
                           pixel_offset, margin);
        /* copy-out as rasterization.py:141-146 */
        const int cx = (tx + 1) * GSO_TILE > result_x ? result_x - tx * GSO_TILE : GSO_TILE;
        const int cy = (ty + 1) * GSO_TILE > result_y ? result_y - ty * GSO_TILE : GSO_TILE;
        for (int lx = 0; lx < cx; lx++)
            for (int ly = 0; ly < cy; ly++) {
                const size_t p = (size_t)(tx * GSO_TILE + lx) * result_y + (ty * GSO_TILE + ly);
                image[3 * p] = chunk[lx][ly][0];
                image[3 * p + 1] = chunk[lx][ly][1];
                image[3 * p + 2] = chunk[lx][ly][2];
                last_contributors[p] = lc[lx][ly];
                if (t_final) t_final[p] = tf[lx][ly];
                if (fragile) fragile[p] = fr[lx][ly];
            }
    }
    return used;
}
