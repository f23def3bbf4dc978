// Piano-roll rasteriser: the scatter at the heart of generate_piano_roll
// (MMGAN_MIDI_DES/datasets.py:29-45) for a batch of MIDI files.
//
//   note_on  (note, step, velocity):  piano_roll[note][step] = velocity;  note_on_time[note] = step
//   note_off (note, step):            off = note_on_time[note];  durations[note][off:step] = step - off
//
// Later messages overwrite earlier ones, so the order of a note's messages matters and nothing else does: rows
// (file, note) are independent.  The host (datasets.py) parses the files, converts message times to one-second steps,
// cuts every file's message list where the reference's loop stops, and hands over each row's messages in file order
// (CSR over file * 128 + note).  One workgroup per file clears the file's two (128, W) planes with coalesced stores,
// then thread `note` replays its row.  Integer work on a few KB per file: bound by launch latency and the planes' bytes.
#include "gdm_common.h"

namespace {

__global__ __launch_bounds__(128) void piano_roll_kernel(const int32_t* __restrict__ row_ptr,
                                                         const int32_t* __restrict__ ev_step,
                                                         const int32_t* __restrict__ ev_vel,     // < 0: note_off
                                                         int W, float* __restrict__ roll, float* __restrict__ dur) {
  const int f = blockIdx.x, note = threadIdx.x;
  float* r = roll + (int64_t)f * 128 * W;
  float* d = dur + (int64_t)f * 128 * W;
  for (int i = threadIdx.x; i < 128 * W; i += 128) { r[i] = 0.f; d[i] = 0.f; }
  __syncthreads();
  r += (int64_t)note * W;
  d += (int64_t)note * W;
  int on_time = 0;                                            // note_on_time = np.zeros(128)
  const int e0 = row_ptr[f * 128 + note], e1 = row_ptr[f * 128 + note + 1];
  for (int e = e0; e < e1; ++e) {
    const int step = ev_step[e], vel = ev_vel[e];
    if (vel >= 0) {
      if (step < W) r[step] = (float)vel;                      // (the host cut the list before a note_on with step >= W)
      on_time = step;
    } else {
      const float len = (float)(step - on_time);
      for (int s = on_time; s < step && s < W; ++s) d[s] = len;     // numpy clips the slice at the array's width
    }
  }
}

}  // namespace

extern "C" int gdm_piano_roll_raster(const int32_t* row_ptr, const int32_t* ev_step, const int32_t* ev_vel, int n_files,
                                     int W, float* roll, float* dur, void* stream) {
  GDM_REQUIRE(row_ptr && roll && dur && n_files > 0 && W > 0, "gdm_piano_roll_raster: bad arguments");
  hipLaunchKernelGGL(piano_roll_kernel, dim3(n_files), dim3(128), 0, (hipStream_t)stream, row_ptr, ev_step, ev_vel, W,
                     roll, dur);
  GDM_LAUNCH_OK("gdm_piano_roll_raster");
  return GDM_OK;
}
