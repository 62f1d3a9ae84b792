/*
 * rtx.h -- C ABI of the MI355X (gfx950) render hot path for sky-5462/RayTracer.
 *
 * The reference has no plugin or FFI seam of its own: its only public method is
 * RayTracer::parseConfigFile (include/RayTracer/RayTracer.h:14) and the hot path is the body
 * of the frame loop in RayTracer::render (src/RayTracer.cpp:237-260).  This header is the seam
 * a maintainer would bind at that loop: the inputs are exactly the reference's private
 * members (RayTracer.h:17-31) in their in-memory layouts, the output is accumulateImg
 * (RayTracer.h:20).  Plain pointers and sizes only; no C++ or torch types.
 *
 * Two libraries implement it:
 *   librtx_hip.so   rtx_*   device side (HIP kernels, gfx950)            raytracer_amd/csrc/rtx_hip.hip
 *   librtx_host.so  rtxh_*  host-side scene preparation (no GPU needed)  raytracer_amd/csrc/rtx_host.cpp
 *
 * Conventions: every function returns 0 on success or a negative rtx_status; the message
 * of the last failure on the calling thread is rtx_last_error()/rtxh_last_error().  No
 * exception crosses the boundary.  The caller owns every host buffer it passes; set_* calls
 * copy synchronously.  One host thread drives a scene at a time.
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rtx_status {
	RTX_OK = 0,
	RTX_ERR_ARG = -1,      /* bad argument, inconsistent scene, size beyond a documented limit */
	RTX_ERR_STATE = -2,    /* call order: e.g. render before the scene is complete */
	RTX_ERR_DEVICE = -3,   /* a HIP call failed; rtx_last_error() holds hipGetErrorString */
	RTX_ERR_MEMORY = -4    /* device or host allocation failed */
} rtx_status;

/* Byte sizes of the reference's records as this ABI expects them. */
#define RTX_TRIANGLE_BYTES 176 /* class Triangle,    include/RayTracer/Triangle.h:6-34  */
#define RTX_NODE_BYTES 48      /* struct LinearNode, include/RayTracer/BVH.h:35-43      */

typedef struct rtx_scene rtx_scene; /* opaque; owns all device memory of one scene on one GPU */

/* Counters of the most recent rtx_render_frames / rtx_primary_hits call. */
typedef struct rtx_stats {
	uint64_t rays;            /* color()-equivalent queries traced (primary + secondary + last-depth occlusion) */
	uint64_t rays_per_level[16];
	uint64_t node_tests;      /* box tests, quantised or exact (only filled by instrumented runs) */
	uint64_t tri_tests;       /* Triangle::hit-equivalent tests   (only filled by instrumented runs) */
	uint64_t shaded_hits;     /* hits that reached the material branch */
	uint64_t texture_fetches; /* texture + skybox texel fetches */
	double trace_ms;          /* sum of HIP-event durations of the traversal kernels */
	double shade_ms;          /* ... of the shade / gather / accumulate kernels */
	double total_ms;          /* HIP-event time of the whole call on the scene's stream */
	uint32_t trace_launches;
	uint32_t chunks;          /* number of pixel batches the frame(s) were split into */
	double trace_ms_per_level[16];
	uint64_t wide_visits;     /* 64-byte nodes of the 4-wide tree fetched (only filled by instrumented runs) */
	uint64_t leaf_fetches;    /* 64-byte triangle records fetched by the wide traversal (instrumented runs)   */
	uint64_t queue_retries;   /* pixel batches rendered again because a queue sized from earlier frames overflowed */
	uint64_t rays_set_aside;  /* rays the wide traversal handed to the binary-tree kernel (rtx_render_frames, rtx_trace_rays) */
	uint64_t rays_untraced;   /* last-level rays (counted in rays) whose value is zero hit or miss: not traversed   */
	uint64_t rays_wave_traced;/* rays whose traversal proved long and was redone by a whole wave (rtx_render_frames only) */
} rtx_stats;

/* ------------------------------------------------------------------ device side (librtx_hip.so) */

const char* rtx_last_error(void);

/* Number of visible HIP devices, or a negative rtx_status. */
int rtx_device_count(void);

/* Creates an empty scene bound to HIP device `device`.  Replaces: construction of the
 * RayTracer object (src/main.cpp:9). */
int rtx_scene_create(rtx_scene** out, int device);
int rtx_scene_destroy(rtx_scene* s);

/* Work is enqueued on `hip_stream` (a hipStream_t; NULL selects a stream the scene owns).
 * Calls still block until their work has finished unless stated otherwise. */
int rtx_scene_set_stream(rtx_scene* s, void* hip_stream);

/* Image size and the scalar members diffuseRayNum, specualrRayNum, maxRecursionDepth,
 * backgroundColor (RayTracer.h:17-18,28-31; parsed at src/RayTracer.cpp:287-332).
 * `seed` keys the counter-based RNG that replaces the reference's thread_local
 * std::mt19937 pair (src/Triangle.cpp:96,122).  Limits: 1 <= max_depth <= 15,
 * 1 <= n_diffuse, n_specular <= 64. */
int rtx_scene_set_params(rtx_scene* s, int width, int height, int max_depth, int n_diffuse, int n_specular,
						 const float background_rgb[3], uint64_t seed);

/* The four Camera members after Camera::setCamera (include/RayTracer/Camera.h:15-18,
 * src/Camera.cpp:4-32), four floats each. */
int rtx_scene_set_camera(rtx_scene* s, const float origin[4], const float left_up_corner[4],
						 const float right_step[4], const float down_step[4]);

/* trianglesArray (RayTracer.h:22): n records of RTX_TRIANGLE_BYTES in the reference's layout. */
int rtx_scene_set_triangles(rtx_scene* s, const void* triangles, uint32_t n);

/* BVH::linearTree (include/RayTracer/BVH.h:53) as produced by BVH::buildTree
 * (src/BVH.cpp:51-160): n records of RTX_NODE_BYTES, root at index 0.  Must be called after
 * rtx_scene_set_triangles; the tree is validated (child indices, one leaf per triangle,
 * depth <= 32) and re-flattened into the device layout. */
int rtx_scene_set_bvh(rtx_scene* s, const void* linear_nodes, uint32_t n);

/* One entry of texturesArray (RayTracer.h:23): `rgb` is Texture::data as the reference
 * holds it after its vertically flipped 3-channel load (src/Texture.cpp:6-7), w*h*3 bytes.
 * Returns the texture id (>= 0) that Triangle::textureIndex refers to. */
int rtx_scene_add_texture(rtx_scene* s, const uint8_t* rgb, int width, int height);

/* Skybox::brightness and the six Skybox::backgroundImg faces (include/RayTracer/Skybox.h:23-24)
 * in the order front, back, left, right, top, bottom (src/Skybox.cpp:13-15), given as
 * texture ids from rtx_scene_add_texture.  Without this call misses return the background
 * colour (src/RayTracer.cpp:154-159). */
int rtx_scene_set_skybox(rtx_scene* s, float brightness, const int texture_ids[6]);

/* Host-computed lookup tables so that no device libm result enters the image:
 * cosTable/sinTable (src/Triangle.cpp:8-21) and powf(c/255.0f, 2.2f) for c = 0..255
 * (src/Texture.cpp:49-51). */
int rtx_scene_set_tables(rtx_scene* s, const float cos512[512], const float sin512[512], const float gamma256[256]);

/* Restricts rendering to the image tiles (64x64 pixels, row-major tile index) first,
 * first+stride, ... (count of them; count < 0 = all that exist).  This is how one GPU of
 * several takes its share of the row loop (src/RayTracer.cpp:237).  Default: all tiles. */
int rtx_scene_set_tiles(rtx_scene* s, int first, int stride, int count);

/* Upper bound in bytes for the per-frame work arena (ray, hit and path-node queues).
 * 0 selects the default (60 % of free device memory, at most 64 GiB). */
int rtx_scene_set_arena_bytes(rtx_scene* s, uint64_t bytes);

/* Uses caller-owned device memory (height*width*4 floats, row-major, the layout of
 * accumulateImg, RayTracer.h:20) as the accumulator instead of the scene's own buffer;
 * NULL switches back. */
int rtx_scene_set_accum_device(rtx_scene* s, void* device_rgba);

/* accumulateImg.fill(Zero) (src/RayTracer.cpp:230). */
int rtx_accum_reset(rtx_scene* s);

/* The loop body of RayTracer::render for frames i = first_frame .. first_frame+n_frames-1
 * (1-based like `i` at src/RayTracer.cpp:234): camera rays, traversal, intersection, bounce
 * loop, accumulateImg(row,col) += sum*0.25f (src/RayTracer.cpp:237-246).  The tone map is
 * rtx_tonemap_read (or the caller's own); the PNG write stays with the caller.
 * With n_frames > 1 the frames are rendered together in merged pixel batches (up to four 1920x1080 frames' worth
 * of pixels per batch, the work arena permitting); every pixel still receives its frames in ascending order, so
 * the accumulator equals that of n_frames calls. */
int rtx_render_frames(rtx_scene* s, uint32_t first_frame, uint32_t n_frames);

/* Copies the accumulator to host memory: height*width*4 floats. */
int rtx_accum_read(rtx_scene* s, float* host_rgba);
/* Uploads an accumulator (resuming from a saved image). */
int rtx_accum_write(rtx_scene* s, const float* host_rgba);

/* The 8-bit tone map of src/RayTracer.cpp:248-257 on the device.  `out8 = clamp(lroundf(powf(acc/frame,
 * 1/2.2f)*255))` is a step function of acc/frame, so the host's own libm is tabulated once
 * (rtxh_tonemap_table) and the device only searches the table: the bytes equal what the host computes.
 * bits[n] ascending float bit patterns of the non-negative breakpoints (bits[0] = 0), values[n] the byte from
 * each breakpoint on; `negative` and `nan` are the bytes for inputs with the sign bit set and for NaN.
 * Inputs beyond 1e15, where the reference's narrowing of lroundf's result to int wraps around, saturate to
 * the last entry. */
int rtx_scene_set_tonemap_table(rtx_scene* s, const uint32_t* bits, const uint8_t* values, int n, uint8_t negative, uint8_t nan);
/* Tone-maps the accumulator after `frame` frames and copies height*width*3 bytes (RGB, row-major, the
 * outputBuffer of src/RayTracer.cpp:257) to host memory. */
int rtx_tonemap_read(rtx_scene* s, int frame, uint8_t* host_rgb8);
/* rtx_render_frames and, for every frame of the call, the image rtx_tonemap_read would give right after that frame
 * (the out_%03d.png of src/RayTracer.cpp:263-269): host_rgb8[k], k = 0..n_frames-1, receives height*width*3 bytes
 * for frame first_frame + k.  The frames still share merged pixel batches -- their queues are filled together, which
 * is what makes several frames cheaper than as many single calls -- and each frame's contribution is added to the
 * accumulator in frame order, with the tone map and the copy of frame k enqueued between the additions of frames k
 * and k + 1.  Needs rtx_scene_set_tonemap_table; page-locked arrays (rtx_host_alloc) are copied at link speed. */
int rtx_render_frames_tonemapped(rtx_scene* s, uint32_t first_frame, uint32_t n_frames, uint8_t* const* host_rgb8);

/* Closest hit of every primary ray (the head of RayTracer::color, src/RayTracer.cpp:137-151,
 * applied to Camera::getRay, src/Camera.cpp:40-49): tri_ids[height*width*4] (reference triangle
 * index or -1), t_alpha_beta[height*width*4*3] (may be NULL).  Sample order: (row*width+col)*4+k.
 * mode 0 = production traversal (ordered, culled); 1 = exhaustive traversal that visits exactly
 * the nodes BVH::hit visits (src/BVH.cpp:162-192), for cross-checking.
 * The device-side arrays stay with the scene from call to call; the host arrays are filled by copies behind the
 * kernels on the scene's stream.  Host arrays from rtx_host_alloc (page-locked) receive them at the speed of the
 * link (a 1920x1080 image: 33 MB of ids, 100 MB of t/alpha/beta); pageable arrays work, several times slower. */
int rtx_primary_hits(rtx_scene* s, int mode, int32_t* tri_ids, float* t_alpha_beta);

/* Page-locked host memory for arrays the library copies out (rtx_primary_hits, rtx_accum_read, rtx_tonemap_read,
 * rtx_sample_radiance): NULL on failure (rtx_last_error says why).  It has no counterpart in the reference, whose
 * images never leave host memory. */
void* rtx_host_alloc(size_t bytes);
int rtx_host_free(void* p);

/* Closest hit of n arbitrary rays (origins/directions as xyz triples), same modes. */
int rtx_trace_rays(rtx_scene* s, int mode, uint32_t n, const float* origins, const float* directions,
				   int32_t* tri_ids, float* t_alpha_beta);

/* rtx_trace_rays for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, code
 * that must not disturb a memory pool).  It has no counterpart in the reference.  The four arrays are DEVICE memory
 * of the scene's GPU, laid out as for rtx_trace_rays (d_t_alpha_beta may be NULL); the call makes no hipMalloc,
 * hipFree or other allocating call: ray records, hit records, the traversal's ray lists and its counters live in a
 * fixed array inside the library (14 MiB per GPU, held until the process ends), which limits one call to
 * RTX_NOALLOC_MAX_RAYS rays; more is RTX_ERR_ARG.  The one word it reads back (the kernels' consistency word)
 * goes into pinned memory the scene obtained when it was created.  The scene needs its triangles and its tree,
 * nothing else.  Work is enqueued on the scene's stream (the default stream if none was set) and the call BLOCKS
 * until it has finished (hipStreamSynchronize), so it cannot be recorded into a HIP graph: a caller under stream
 * capture has to end the capture first.  All scenes of one GPU share that GPU's array: calls that overlap on one
 * GPU race.  No timing and no test counters are taken (rtx_get_stats reports the ray count only). */
#define RTX_NOALLOC_MAX_RAYS (1u << 18)
int rtx_trace_rays_noalloc(rtx_scene* s, int mode, uint32_t n, const float* d_origins, const float* d_directions,
						   int32_t* d_tri_ids, float* d_t_alpha_beta);

/* Per-sample radiance of one frame before accumulation, i.e. color(0, ray) for every
 * primary ray: out[height*width*4][4].  Tiles outside the scene's tile set are left untouched. */
int rtx_sample_radiance(rtx_scene* s, uint32_t frame, float* host_out);

/* Several GPUs in one process: the exchange step after the row loop that rtx_scene_set_tiles split
 * (src/RayTracer.cpp:237; the reference's loop writes one shared accumulateImg, here every GPU holds its tile
 * share in an image of its own).  scenes[r], r = 0..n-1, have the same image size and hold the tile share
 * (r, n) -- rtx_scene_set_tiles(scenes[r], r, n, -1).  The accumulator tiles of scenes 1..n-1 are packed on their
 * GPUs, sent to scenes[0]'s GPU and written into scenes[0]'s accumulator, which then is the whole image for
 * rtx_tonemap_read / rtx_accum_read.  The other accumulators are left as they are.
 * Transport: the scenes live on n different GPUs -- RCCL, one ncclSend/ncclRecv pair per peer inside one group,
 * i.e. one message per xGMI link, all at once; communicators are created on the first call (ncclCommInitAll over
 * the scenes' devices) and kept for later calls with the same devices; librccl.so is opened at the first call, a
 * process that never gets here does not need it -- or all on the same GPU (several worker threads sharing one
 * device): then a message is a device-to-device copy.  Anything in between is RTX_ERR_ARG.  With n = 1 the single
 * share is packed, sent to itself through RCCL and unpacked (what a test on one GPU can exercise of that path).
 * Blocks until the image is complete.  One thread at a time. */
int rtx_gather_tiles(rtx_scene* const* scenes, int n);

/* The same exchange for dynamically balanced frames, where which scene renders which tile changes from frame to
 * frame (worker threads pulling tile batches from a shared counter: rtx_scene_set_tiles(s, first, 1, count) +
 * rtx_render_frames per batch).  Every scene keeps a log of the tiles it has rendered since the log was last
 * cleared; the call moves exactly those tiles: a sender's logged tiles are packed and left ZERO, scenes[0] ADDS
 * what it receives to its accumulator (x = x + contribution: bit for bit what rendering the tile into scenes[0]'s
 * accumulator would have given, because a zeroed accumulator plus one frame's contribution is that contribution
 * exactly), and all logs are cleared.  Called once per frame, it keeps scenes[0]'s accumulator equal to the
 * single-GPU one; the accumulators of the other scenes are frame buffers that are empty between frames.  A tile
 * logged by two scenes is RTX_ERR_ARG.  Transport and blocking as for rtx_gather_tiles. */
int rtx_exchange_tiles(rtx_scene* const* scenes, int n);

/* The log rtx_exchange_tiles works from: writes up to `capacity` tile numbers (ascending) and returns how many the
 * log holds.  rtx_render_frames adds the tiles of the scene's share to it; rtx_exchange_tiles, rtx_accum_reset and
 * rtx_scene_tile_log_clear empty it. */
int rtx_scene_tile_log(rtx_scene* s, int* tiles, int capacity);
int rtx_scene_tile_log_clear(rtx_scene* s);

/* When enabled, traversal kernels count box and triangle tests (slower). */
int rtx_scene_set_instrumented(rtx_scene* s, int enabled);
int rtx_get_stats(rtx_scene* s, rtx_stats* out);

/* ------------------------------------------------------------------ host side (librtx_host.so) */

const char* rtxh_last_error(void);

/* Camera::setCamera (src/Camera.cpp:4-32).  out16 = origin, leftUpCorner, rightStep, downStep. */
int rtxh_camera(const float eye[3], const float view_point[3], float focal, float rotate_deg, int width, int height,
				float out16[16]);

/* BVH::buildTree (src/BVH.cpp:51-160): writes the node records (capacity in records;
 * 3n+1 always suffices: n leaves plus at most 2n-1 interior nodes) and returns the node count, or a
 * negative rtx_status. */
int rtxh_build_bvh(const void* triangles, uint32_t n, void* nodes_out, uint32_t capacity);

/* RayTracer::addTriangle (src/RayTracer.cpp:108-134): fills one RTX_TRIANGLE_BYTES record. */
int rtxh_make_triangle(const float v0[3], const float v1[3], const float v2[3], const float normal_side[3],
					   const float color[3], int is_metal, int is_light_emitting, int is_transparent,
					   float specular_roughness, float refractive_index, void* triangle_out);

/* The static tables of src/Triangle.cpp:8-21 and the texel decode of src/Texture.cpp:49-51. */
int rtxh_tables(float cos512[512], float sin512[512], float gamma256[256]);

/* Tone map of src/RayTracer.cpp:248-257: rgb8[npix*3] from accum[npix*4] after `frame` frames. */
int rtxh_tonemap(const float* accum_rgba, int frame, size_t npix, uint8_t* rgb8);
/* The same function as a table for rtx_scene_set_tonemap_table: up to `cap` breakpoints (1024 is plenty)
 * found by bisection over the float bit patterns with this host's powf/lroundf; special[0] = byte for negative
 * inputs, special[1] = byte for NaN. */
int rtxh_tonemap_table(uint32_t* bits, uint8_t* values, int cap, int* n_out, uint8_t special[2]);

/* Texture::Texture(path) (src/Texture.cpp:4-8): decodes an image file (JPEG, PNG, BMP, GIF, PSD, PIC, HDR, TGA,
 * PPM/PGM: the containers of the reference's loader) to 8-bit RGB, bottom row first as the reference keeps it.  Writes width and height; copies the pixels when `rgb` is not
 * null and `capacity` >= width*height*3.  Returns the byte count needed, or a negative rtx_status. */
long long rtxh_load_texture(const char* path, int* width, int* height, uint8_t* rgb, size_t capacity);

/* RayTracer::loadModel (src/RayTracer.cpp:20-106) without its texture file, which the caller loads with
 * rtxh_load_texture: reads a mesh (Wavefront OBJ with the Kd of its .mtl, Stanford PLY, STL, OFF; polygons fanned),
 * places it (position * scale + origin) and fills RTX_TRIANGLE_BYTES records with the given material, the plane
 * normal of src/RayTracer.cpp:96 and the file's vertex normals (the face normal when it has none).
 * texture_index: the index the model's texture has in the scene, or -1; like the reference (:65-66) it is kept only
 * when the mesh has texture coordinates, and *textured (may be null) says whether it was.  override_color: three
 * floats or null (then the file's diffuse colour; 0.6 grey when it names none).  With triangles_out null only the
 * triangle count is returned; otherwise `capacity` (in records) must hold it.  Returns the count or a negative
 * rtx_status ("Can't load model in <path>", the reference's message, in rtxh_last_error). */
long long rtxh_load_model(const char* model_path, const float origin[3], float scale, int is_metal, int is_light_emitting,
						  int is_transparent, float specular_roughness, float refractive_index, int texture_index,
						  const float* override_color, void* triangles_out, size_t capacity, int* textured);

/* Whole drop-in: parse `config_path` with the grammar of RayTracer::parseConfigFile
 * (src/RayTracer.cpp:284-518), render on `n_devices` GPUs (device ids 0..n-1) and write
 * out_%03d.png into `out_dir` after every frame like src/RayTracer.cpp:263-271.
 * Implemented in librtx_host.so on top of librtx_hip.so, which it loads at run time. */
int rtxh_run_config(const char* config_path, const char* out_dir, int n_devices);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
