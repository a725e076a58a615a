/* libmistitch: C ABI of the MI355X (gfx950) registration-and-fusion hot path.
 *
 * The reference (maragall/image-stitcher) is pure Python and has no FFI of its
 * own; the seam this library slots under is the set of Python functions cited
 * beside each entry point (paths relative to the reference checkout).  A
 * ctypes binding calls these functions with device pointers obtained from
 * torch tensors (`tensor.data_ptr()`).
 *
 * Conventions
 *   - every array argument is a DEVICE pointer unless the comment says host;
 *   - the library never frees or retains caller memory; it owns only the
 *     context's plans and workspaces;
 *   - work is enqueued on the context's stream and is asynchronous with
 *     respect to the host unless stated otherwise;
 *   - every function returns MISTITCH_OK (0) or a negative error code, and
 *     mistitch_last_error() then describes the failure (per calling thread);
 *   - one call at a time per context; different contexts are independent.
 */
#ifndef MISTITCH_H
#define MISTITCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MISTITCH_OK 0
#define MISTITCH_ERR_ARG (-1)         /* invalid argument (NULL, negative size, bad enum) */
#define MISTITCH_ERR_UNSUPPORTED (-2) /* shape outside what the kernels are built for     */
#define MISTITCH_ERR_HIP (-3)         /* a HIP runtime call failed                        */
#define MISTITCH_ERR_NOMEM (-4)       /* device allocation failed                         */

/* per-pair status written by mistitch_register_pairs */
#define MISTITCH_PAIR_OK 0
#define MISTITCH_PAIR_FEATURELESS 1  /* a strip failed the variance test: reference returns None */
#define MISTITCH_PAIR_NO_CANDIDATE 2 /* no candidate scored: row holds (-inf, off_y, off_x)      */

/* flags */
#define MISTITCH_FFT_F64 0u /* complex128 spectra, as the reference computes them (default) */
#define MISTITCH_FFT_F32 1u /* complex64 spectra: half the traffic, PCM differs at ~1e-7     */
#define MISTITCH_REG_SUBPIXEL 2u /* mistitch_register_pairs: add the upsampled-DFT estimate    */
#define MISTITCH_REG_PEAKS10 4u  /* mistitch_register_pairs: search the 10 best peaks (160 candidates) */
#define MISTITCH_REG_CONSENSUS 16u /* mistitch_register_pairs: the consensus stage over those 160 candidates (implies
                                      MISTITCH_REG_PEAKS10), see mistitch_register_pairs_candidates */
#define MISTITCH_REG_WINDOW_HANN 8u /* registration, mistitch_pcm*, mistitch_subpixel_shift: multiply both strips
                                       by a separable symmetric Hann window (numpy.hanning along either axis, in
                                       float64) before the transforms; the candidate NCCs stay on raw pixels.
                                       OFF by default: the reference applies no window
                                       (_translation_computation.py:308-313), so its results need this flag unset;
                                       smooth scenes whose unwindowed correlation peaks at leakage near (0, 0)
                                       register with it.  No counterpart in the reference. */

/* fusion modes */
#define MISTITCH_FUSE_OVERWRITE 0 /* last writer wins, the reference behaviour */
#define MISTITCH_FUSE_FEATHER 1   /* separable linear-ramp weighted mean (no reference counterpart) */

/* flatfield element types */
#define MISTITCH_FF_NONE 0
#define MISTITCH_FF_F32 1
#define MISTITCH_FF_F64 2

typedef struct mistitch_ctx mistitch_ctx;

int mistitch_version(void);

/* Message for the most recent error on the calling thread ("" if none). */
const char* mistitch_last_error(void);

/* Create a context on `device`.  `stream` is a hipStream_t (NULL = the legacy
 * default stream).  Replaces the reference's global backend singleton
 * (image_stitcher/registration/tile_registration.py:72-97) for this path. */
int mistitch_create(int device, void* stream, mistitch_ctx** out);
int mistitch_destroy(mistitch_ctx* ctx);

/* Pairs are processed in passes of `pairs_per_pass` (>=1) to bound the FFT
 * workspace (about 19 MB per pair at edge width 256 with complex128 spectra).
 * Default 128. */
int mistitch_set_pair_chunk(mistitch_ctx* ctx, int pairs_per_pass);

/* A call with more pairs than one pass stages the edge strips of ALL its pairs first (one
 * launch) and sums the candidate overlaps once after the last pass, when that staging buffer
 * (2 strips per pair: 2 MiB at edge width 256 on 2048-pixel tiles) stays within `bytes`;
 * otherwise strips are staged and candidates summed pass by pass, with the same results.
 * Default 4 GiB; 0 = always pass by pass.  No counterpart in the reference, whose pair loop
 * is serial. */
int mistitch_set_staging_limit(mistitch_ctx* ctx, int64_t bytes);

/* Pre-size all workspaces so that later mistitch_register_pairs calls with
 * shapes up to these do not allocate, copy tables or change kernel attributes (and can
 * therefore be captured into a HIP graph): sizes the buffers, then makes one registration
 * call of `max_pairs` pairs on two blank tiles with `flags` and waits for it. */
int mistitch_reserve_registration(mistitch_ctx* ctx, int tile_h, int tile_w, int edge_width,
                                  int max_pairs, unsigned flags);

/* Pairwise registration of neighbouring tiles.
 * Replaces the serial pair loop of process_neighborhood_batch
 * (tile_registration.py:310-378) and, per pair, compute_single_translation
 * (tile_registration.py:527-586): featureless test, phase correlation, peak
 * ranking, 16-candidate NCC search over the two best peaks.  That is what the
 * reference computes where scikit-image is not installed (its import guard,
 * _translation_computation.py:13-22, then skips "Method 1" and "Method 2" of
 * interpret_translation_ransac, :1197-1272); MISTITCH_REG_CONSENSUS and
 * MISTITCH_REG_SUBPIXEL add those two methods.
 *
 *   tiles      [n_tiles] images of tile_h x tile_w uint16, row-major, image t at
 *              tiles + t * tile_pitch (elements)
 *   pair_idx   [n_pairs][2] int32: (first, second); second is the right ("left"
 *              pair, dir 0) or lower ("top" pair, dir 1) neighbour
 *   pair_dir   [n_pairs] uint8: 0 = left, 1 = top
 *   out        [n_pairs][3] float64: ncc, y, x of the second tile's origin in
 *              the first tile's frame; NaN row when status is FEATURELESS
 *   status     [n_pairs] int8: MISTITCH_PAIR_*
 *
 * With MISTITCH_REG_SUBPIXEL in `flags` the upsampled-DFT estimate (see
 * mistitch_subpixel_shift) is scored by NCC at its rounded position and competes
 * with the discrete search as "Method 2" does in the reference when scikit-image
 * is installed (_translation_computation.py:1232-1320): it wins unless strictly
 * worse, and y, x are then fractional.
 *
 * Strips whose sides are powers of two (long side 8..4096, short side 8..1024)
 * take the Hermitian radix-16 path; with a power-of-two long side and any short
 * side up to 1024 only the column transforms become chirp-z ones (complex128
 * whatever `flags` says, about 2.5 times slower); any other side lengths up to
 * 4096 take chirp-z transforms on both axes (complex128, several times slower).
 */
int mistitch_register_pairs(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h,
                            int tile_w, int64_t tile_pitch, const int32_t* pair_idx,
                            const uint8_t* pair_dir, int n_pairs, int edge_width, unsigned flags,
                            double* out, int8_t* status);

/* mistitch_register_pairs on RAW tiles with the flatfield of the registration channel
 * applied on the way: the edge strips are extracted, corrected (trunc(clip(v / ff)), one
 * flatfield plane of tile_h x tile_w, ff_type MISTITCH_FF_F32 / _F64) and staged in one
 * pass, so only the quarter of every tile that lies in strips is ever corrected.  Same
 * results as correcting whole tiles first, which is what the reference does when it
 * loads them (load_batch_images -> apply_flatfield_to_image,
 * tile_registration.py:2184-2279, 2142-2181) before process_neighborhood_batch.        */
int mistitch_register_pairs_flatfield(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h,
                                      int tile_w, int64_t tile_pitch, const void* flatfield, int ff_type,
                                      const int32_t* pair_idx, const uint8_t* pair_dir, int n_pairs,
                                      int edge_width, unsigned flags, double* out, int8_t* status);

/* Flag MISTITCH_REG_PEAKS10: the candidate search walks the 16 signed interpretations of the TEN
 * largest PCM peaks instead of two: the candidate set of the reference's "Method 1"
 * (generate_translation_candidates(n=10), _translation_computation.py:876-994), whose
 * best-NCC candidate is what that method returns whenever its consensus step declines
 * (ransac_translation_estimation, :1019-1038 and :1112-1118).  The
 * result is the first candidate of the walk that no other beats; with MISTITCH_REG_SUBPIXEL
 * as well the subpixel estimate only wins when strictly better (Method 1 is listed first).
 *
 * Flag MISTITCH_REG_CONSENSUS (implies MISTITCH_REG_PEAKS10): the consensus step of "Method 1" as
 * the reference executes it.  ransac_translation_estimation hands skimage.measure.ransac a model
 * class whose estimate(self, data, *args) unpacks `src, dst = data` (:1053-1055); ransac calls
 * estimate(*samples) with a 3-row sample (min_samples = 3, :74), the unpacking raises, and the
 * reference's own `except` branch (:1085-1107) decides, deterministically: pairwise Euclidean
 * distances of the candidates of finite NCC, neighbours within residual_threshold = 2.0 (the
 * candidate itself included), the FIRST candidate with the most neighbours; when those neighbours are
 * fewer than min_inlier_ratio = 0.3 of the candidates (:1110-1118), or the candidates fewer than 3
 * (:1030-1038), the first candidate of the largest NCC instead.  Its shift, scored at that
 * position (:1217-1227), heads the result list, then the subpixel estimate (MISTITCH_REG_SUBPIXEL),
 * then the two-peak discrete search (:1276-1280); the first entry of the largest NCC is returned
 * (:1313).  A scikit-image whose ransac does NOT raise on that model class would draw random samples
 * from an unseeded generator instead: that behaviour is not reproduced (parity unpinned there).
 *
 * mistitch_register_pairs_candidates: mistitch_register_pairs (flatfield == NULL, ff_type
 * MISTITCH_FF_NONE) or mistitch_register_pairs_flatfield, plus the table the walk scored:
 *   table      [n_pairs][n_cand][3] float64, n_cand = 32 (160 with MISTITCH_REG_PEAKS10):
 *              y, x (strip relative, before the strip offset is added) and ncc of every
 *              candidate in the reference's evaluation order; ncc is NaN where the
 *              reference does not score the candidate (empty overlap, fewer peaks than
 *              asked for) and -inf for an overlap of fewer than 25 pixels; rows of
 *              featureless pairs are NaN                                                  */
int mistitch_register_pairs_candidates(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h,
                                       int tile_w, int64_t tile_pitch, const void* flatfield, int ff_type,
                                       const int32_t* pair_idx, const uint8_t* pair_dir, int n_pairs,
                                       int edge_width, unsigned flags, double* out, int8_t* status,
                                       double* table);

/* Integer hill climb of the strip NCC around given start shifts: from start_yx the
 * best of the 3 x 3 integer neighbours (first of equal maxima, row-major) is taken
 * while it strictly improves, within +-r of the start and the tile extent, at most
 * 100 moves.  Replaces the pair loop of refine_translations and
 * find_local_max_integer_constrained (registration/_constrained_refinement.py:
 * 140-307, 62-138).  Arguments as mistitch_register_pairs, plus
 *   start_yx   [n_pairs][2] int32: start shift of the second tile in the first tile's frame
 *   out        [n_pairs][3] float64: ncc, y, x; (NaN, start) where the reference keeps
 *              the start (start outside its limits or not scoring)
 *   status     [n_pairs] int8: MISTITCH_PAIR_OK, or MISTITCH_PAIR_FEATURELESS for a pair
 *              with a tile index out of range                                          */
int mistitch_refine_pairs(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h, int tile_w,
                          int64_t tile_pitch, const int32_t* pair_idx, const uint8_t* pair_dir,
                          int n_pairs, int edge_width, const int32_t* start_yx, double r, double* out,
                          int8_t* status);

/* Subpixel shift between two equal-shape uint16 strips by upsampled DFT: the
 * coarse shift is the largest |PCM| value, refined on a ceil(1.5 u) square grid
 * of spacing 1/u.  Replaces skimage.registration.phase_cross_correlation(a, b,
 * upsample_factor=u, normalization='phase') as the reference calls it
 * (_translation_computation.py:1233-1272).  The grid values are not summed one by one:
 * the correlation is evaluated at 32 x 32 Chebyshev nodes of the grid's span and
 * interpolated to the grid, which for a function of its bandwidth is exact to rounding
 * error (the argmax on the grid is that of the direct sums).  Any strip sides up to 4096
 * (strips of other than power-of-two sides are transformed in float64 whatever the flags
 * say); 2 <= u <= 106.
 *   out_shift  [2] float64: (dy, dx)                                               */
int mistitch_subpixel_shift(mistitch_ctx* ctx, const uint16_t* strip1, int64_t pitch1,
                            const uint16_t* strip2, int64_t pitch2, int h, int w, unsigned flags,
                            int upsample_factor, double* out_shift);

/* Upsample factor used by mistitch_register_pairs with MISTITCH_REG_SUBPIXEL
 * (default 100, SubpixelConfig.upsample_factor at tile_registration.py:574). */
int mistitch_set_upsample_factor(mistitch_ctx* ctx, int upsample_factor);

/* The largest value of the acquisition's pixel type (default 65535).  The reference takes the dtype of the first image
 * (parameters.py:685) and clips flatfield-corrected pixels to that type's range (stitcher.py:260-267,
 * tile_registration.py:2177-2180): tiles of an 8-bit camera, widened to uint16 for this library, are corrected with
 * pixel_max = 255.  Applies to every later call on the context that divides by a flatfield.                       */
int mistitch_set_pixel_max(mistitch_ctx* ctx, int pixel_max);

/* Peak correlation matrix of two equal-shape uint16 strips.
 * Replaces pcm() (registration/_translation_computation.py:282-340).
 *   strip1/2   h x w uint16 with row pitches pitch1/pitch2 (elements)
 *   out_pcm    [h][w] float64 (always float64, whatever the FFT precision)     */
int mistitch_pcm(mistitch_ctx* ctx, const uint16_t* strip1, int64_t pitch1, const uint16_t* strip2,
                 int64_t pitch2, int h, int w, unsigned flags, double* out_pcm);

/* The k (1..16) largest PCM entries of two strips in descending order; equal
 * values are ordered by descending flat index.  Replaces
 * multi_peak_max(pcm(a, b), max_peaks=k) (_translation_computation.py:342-441).
 *   rows, cols [k] int32;  vals [k] float64                                     */
int mistitch_pcm_topk(mistitch_ctx* ctx, const uint16_t* strip1, int64_t pitch1,
                      const uint16_t* strip2, int64_t pitch2, int h, int w, unsigned flags, int k,
                      int32_t* rows, int32_t* cols, double* vals);

/* Pearson correlation of two h x w uint16 patches, exact integer moments.
 * Replaces ncc(image1, image2, min_overlap_pixels) (_translation_computation.py:535-616):
 * -inf for an empty patch or when h*w < min_overlap_pixels (the reference's default is 25).
 *   out        [1] float64                                                      */
int mistitch_ncc(mistitch_ctx* ctx, const uint16_t* patch1, int64_t pitch1, const uint16_t* patch2,
                 int64_t pitch2, int h, int w, int min_overlap_pixels, double* out);

/* The strip variance test of is_edge_featureless (tile_registration.py:497-525):
 * population variance of every 16th sample in both directions.
 *   out_var    [1] float64                                                      */
int mistitch_edge_variance(mistitch_ctx* ctx, const uint16_t* strip, int64_t pitch, int h, int w,
                           double* out_var);

/* out = trunc(clip(in / flatfield, 0, 65535)) per pixel, n_tiles images sharing
 * one flatfield plane.  Replaces apply_flatfield_to_image
 * (tile_registration.py:2142-2181) and Stitcher.apply_flatfield_correction
 * (stitcher.py:255-270).  ff_type: MISTITCH_FF_F32 or MISTITCH_FF_F64.
 * in and out may alias.                                                         */
int mistitch_apply_flatfield(mistitch_ctx* ctx, const uint16_t* in, const void* flatfield,
                             int ff_type, uint16_t* out, int n_tiles, int tile_h, int tile_w);

/* Fuse tiles into a mosaic.  Replaces the tile loop of Stitcher.stitch_region
 * (stitcher.py:388-405) with place_tile / place_single_channel_tile
 * (stitcher.py:208-253, 272-311).
 *
 *   tiles      [n_tiles] images tile_h x tile_w uint16, image t at tiles + t*tile_pitch
 *   tile_yx    [n_tiles][2] int32 mosaic position (y, x) of each tile's origin; a tile that sticks out of
 *              the mosaic on any side (negative coordinates included: a row band fused as a mosaic of
 *              its own sees the tiles that straddle its first row that way) is clipped
 *   tile_plane [n_tiles] int32 destination plane (channel * Z + z) of each tile
 *   n_planes   planes in the mosaic; planes are mosaic_h x mosaic_w, contiguous
 *   tile array order IS the placement order: a later tile overwrites an earlier one
 *   flatfields [n_ff_planes] planes tile_h x tile_w (ff_type), or NULL
 *   tile_ff    [n_tiles] int32 flatfield plane per tile (-1 = none), or NULL
 *   mosaic     [n_planes][mosaic_h][mosaic_w] uint16, fully written for rows
 *              [row0, row1) of every plane (uncovered pixels = 0)
 *   mode       MISTITCH_FUSE_OVERWRITE | MISTITCH_FUSE_FEATHER
 */
int mistitch_fuse(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h, int tile_w,
                  int64_t tile_pitch, const int32_t* tile_yx, const int32_t* tile_plane,
                  const void* flatfields, int ff_type, const int32_t* tile_ff, uint16_t* mosaic,
                  int n_planes, int mosaic_h, int mosaic_w, int row0, int row1, int mode);

/* mistitch_fuse for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.
 * Same arguments and results as mistitch_fuse, but the per-block tile lists live in a 16 MiB
 * array that is part of the library's code object (one copy per GPU, held until the process
 * ends) instead of the context's grow-only buffer: the call makes no hipMalloc / hipFree.
 * Limit: (number of 32-row x 2048-pixel output blocks of rows [row0, row1) over all planes)
 * * 132 + 16 bytes <= 16 MiB, i.e. about 127 000 blocks or 8.3 gigapixels per call (the
 * 4 x 37 070 x 37 071 mosaic of a 20 x 20 x 4 acquisition needs 11.7 MB); beyond it the call
 * fails with MISTITCH_ERR_UNSUPPORTED and writes nothing - fuse in row bands then.
 * All contexts of one GPU share that array: calls that overlap on one GPU (different streams
 * or threads) race; calls on one stream are safe back to back.                             */
int mistitch_fuse_static(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h, int tile_w,
                         int64_t tile_pitch, const int32_t* tile_yx, const int32_t* tile_plane,
                         const void* flatfields, int ff_type, const int32_t* tile_ff, uint16_t* mosaic,
                         int n_planes, int mosaic_h, int mosaic_w, int row0, int row1, int mode);

/* mistitch_fuse with the mosaic born in the chunk order of the array writer: `chunks` is
 * [n_planes][ncy][ncx][chunk_h][chunk_w] uint16 with ncy = ceil(mosaic_h / chunk_h), ncx = ceil(mosaic_w / chunk_w),
 * every [cy][cx] one whole C-ordered chunk of a Zarr array with chunks (1, 1, 1, chunk_h, chunk_w) (the reference's
 * chunk shape is one tile, parameters.py:695-701), zero where it sticks out of the mosaic: the bytes
 * mistitch_chunk_layout makes of the plain mosaic, without writing and re-reading that mosaic.  Replaces the tile loop
 * of Stitcher.stitch_region (stitcher.py:388-405) together with the rechunking that da.store does on the host for
 * save_region_ome_zarr (stitcher.py:575-594).  Rows [row0, row1) of the mosaic are written; the band that reaches
 * mosaic_h also writes the zero rows below it in the last chunks.  chunk_w must be a multiple of 8 and `chunks`
 * 16-byte aligned (MISTITCH_ERR_UNSUPPORTED / _ARG otherwise: fuse plainly and call mistitch_chunk_layout).
 * All other arguments as mistitch_fuse.                                                                          */
int mistitch_fuse_chunked(mistitch_ctx* ctx, const uint16_t* tiles, int n_tiles, int tile_h, int tile_w,
                          int64_t tile_pitch, const int32_t* tile_yx, const int32_t* tile_plane,
                          const void* flatfields, int ff_type, const int32_t* tile_ff, uint16_t* chunks,
                          int n_planes, int mosaic_h, int mosaic_w, int chunk_h, int chunk_w, int row0, int row1,
                          int mode);

/* Per-pixel maximum over `n` images of h*w uint16 (image i at in + i*pitch).
 * Replaces Stitcher.compute_mip (stitcher.py:102-121).                           */
int mistitch_mip(mistitch_ctx* ctx, const uint16_t* in, int n, int64_t pitch, int64_t n_pixels,
                 uint16_t* out);

/* Resolution pyramid of a mosaic: levels[l] (l = 1 .. n_levels-1, device pointers in
 * a HOST array; levels[0] is ignored) receives [n_planes][h >> l][w >> l] float64, the
 * mean of every 2^l x 2^l block, blocks that stick out of the image dropped.  All
 * levels are produced from one read of the image.  Replaces the per-level
 * dask.array.coarsen(np.mean, ..., trim_excess=True) of Stitcher.generate_pyramid
 * (stitcher.py:652-668), whose levels are float64 too.  n_levels <= 7.            */
int mistitch_pyramid(mistitch_ctx* ctx, const uint16_t* image, int n_planes, int h, int w,
                     int n_levels, double* const* levels);

/* mistitch_pyramid of a mosaic that was born in chunk order (mistitch_fuse_chunked): `chunks` is
 * [n_planes][ceil(h / chunk_h)][ceil(w / chunk_w)][chunk_h][chunk_w]; the levels come out as mistitch_pyramid's,
 * row-major planes of float64.  chunk_w must be a multiple of 8.                                          */
int mistitch_pyramid_chunked(mistitch_ctx* ctx, const uint16_t* chunks, int n_planes, int h, int w, int chunk_h,
                             int chunk_w, int n_levels, double* const* levels);

/* mistitch_pyramid with every LEVEL written in chunk order as well: levels[l] receives
 * [n_planes][ceil((h >> l) / chunk_h)][ceil((w >> l) / chunk_w)][chunk_h][chunk_w] float64, what sticks out of the
 * level zero - the buffers the reference's per-level da.store hands to its chunk files (stitcher.py:526-594) without a
 * layout pass over the float64 levels (the first one of a 4 x 37k x 37k mosaic is 11 GB).  `image` is row-major planes,
 * or the chunk-ordered mosaic of mistitch_fuse_chunked when image_is_chunked is non-zero (same chunk shape, chunk_w a
 * multiple of 8 then).                                                                                        */
int mistitch_pyramid_to_chunks(mistitch_ctx* ctx, const uint16_t* image, int n_planes, int h, int w, int chunk_h,
                               int chunk_w, int image_is_chunked, int n_levels, double* const* levels);

/* The planes of an image, [n_planes][h][w] of uint16 (elem_size 2) or float64 (elem_size 8:
 * the pyramid levels), laid out as the chunks a Zarr array with chunks (1, 1, 1, chunk_h,
 * chunk_w) stores them: out[plane][cy][cx][chunk_h][chunk_w], cy < ceil(h / chunk_h),
 * cx < ceil(w / chunk_w), every chunk whole and C-ordered, the part of an edge chunk that
 * sticks out of the image zero (the array's fill value).  Each chunk of `out` is then the
 * exact byte content of one uncompressed chunk file (little-endian), or the input block of
 * its compressor.  Serves the writer of Stitcher.save_region_ome_zarr (stitcher.py:526-594;
 * chunk shape from parameters.py:695-701), which the reference leaves to dask's
 * `da.store` rechunking on the host.                                                */
int mistitch_chunk_layout(mistitch_ctx* ctx, const void* image, int elem_size, int n_planes, int h, int w,
                          int chunk_h, int chunk_w, void* out);

/* Blosc's BITSHUFFLE filter on the device: the first stage of the reference's default chunk compressor,
 * Blosc(cname='zstd', clevel=5, shuffle=Blosc.BITSHUFFLE) (stitcher.py:536, :549), which numcodecs applies to every
 * chunk on the host.  The buffer is taken in blocks of block_bytes (the codec's block size; the last block may be
 * shorter); inside a block the n = floor(elements / 8) * 8 leading elements are transposed as a bit matrix,
 *   out[(8 b + k) * (n / 8) + g] = bit k of byte b of elements 8 g .. 8 g + 7, element 8 g + i in bit i,
 * and the bytes behind them are copied (bitshuffle's bshuf_trans_bit_elem as c-blosc drives it).  The entropy coder
 * itself (zstd) and the Blosc container stay with the caller's codec library: chunks filtered here are compressed with
 * shuffle off.  Neither numcodecs nor c-blosc is available to check against: tested against a NumPy statement of the
 * definition above only.
 *   in, out    n_bytes each, device memory, not aliased;  elem_size 1, 2, 4 or 8                                  */
int mistitch_bitshuffle(mistitch_ctx* ctx, const void* in, void* out, int64_t n_bytes, int elem_size,
                        int64_t block_bytes);

/* Per-kernel timing with HIP events recorded on the context's stream around
 * every launch (measurement aid; no counterpart in the reference).  Enable,
 * run any entry points, then collect: kernel names come back '\n'-separated in
 * first-launch order with the summed milliseconds and launch counts.  Both
 * calls synchronise the stream.                                                   */
int mistitch_profile_enable(mistitch_ctx* ctx, int on);
int mistitch_profile_collect(mistitch_ctx* ctx, char* names, size_t names_len, double* total_ms,
                             int* launches, int max_entries, int* n_entries);

/* Block the calling thread until the context's stream has drained. */
int mistitch_synchronize(mistitch_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MISTITCH_H */
