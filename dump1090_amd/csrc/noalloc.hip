// modes_convert_device_noalloc(): the stand-alone converter for callers that cannot allocate (code under HIP graph
// capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
//
// It runs the same kernels as modes_convert_device() (kernels.hip: convert_kernel, dc_zscan_kernel, dc_mag_kernel,
// nodc_mag_kernel, power_seq_kernel); what that call keeps in hipMalloc'ed scratch -- the as-written uc8 table, the
// u64 power accumulator, the z1 sequences and mag_sq of the float converters -- lives here in ONE file-scope device
// array per GPU, and the results that call hands back through host memory go to device memory the caller passes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

#include "../../include/modes_hip.h"
#include "common.h"
#include "kernels.h"
#include "tables.h"

namespace modes { int set_error (int code, const char *fmt, ...); }
using namespace modes;

namespace {

// Layout of the scratch array (bytes):
//   [0, 512)        mag_lut as written, 256 x u16 (filled once at set-up)
//   [512, 520)      u64 sum of squared magnitudes (convert_uc8_nodc_power)          -- cleared by every call
//   [528, 544)      float z_final[2], power, pad                                      -- written by every call that reads it
//   [1024, ...)     three float arrays of MODES_NOALLOC_MAX_SAMPLES: z1_I, z1_Q sequences, mag_sq
constexpr size_t kOffLut = 0, kOffPower = 512, kOffSmall = 528, kOffFloats = 1024;
constexpr size_t kMaxN = MODES_NOALLOC_MAX_SAMPLES;
constexpr size_t kScratchBytes = kOffFloats + 3 * kMaxN * sizeof (float);

static __device__ __align__ (256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_mu;
unsigned char *g_ptr[kMaxDevices];          // device address of `scratch` on each GPU (each GPU holds its own copy)

// out_power / final state to the caller's device memory, in the types the host entry point reports them
__global__ void noalloc_result_kernel (int conv, int dc, float z1_I_in, float z1_Q_in, const unsigned long long *power_u64,
                                       const float *small, modes_convert_result *out)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  modes_convert_result r;
  r.out_power = 0.0;
  r.z1_I = z1_I_in; r.z1_Q = z1_Q_in;
  if (conv == kConvUc8LutPower) r.out_power = double (*power_u64) / 65535.0 / 65535.0;     // ref: src/convert.c:107-108
  else if (conv == kConvUc8Generic || conv == kConvSc16Generic || conv == kConvSc16Q11Generic) {
    r.out_power = double (small[2]);                                                        // ref: src/convert.c:153-154
    if (dc) { r.z1_I = small[0]; r.z1_Q = small[1]; }
  }
  *out = r;
}

// once per device, with that device current: the address of this GPU's copy of the array, and what must be in it
int setup_device (int dev, hipStream_t stream)
{
  if (g_ptr[dev]) return MODES_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing (stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
    return set_error (MODES_E_ARG, "modes_convert_device_noalloc: the first call on a device sets up its tables and must "
                                   "not be made while the stream is being captured");
  unsigned char *ptr = nullptr;
  hipError_t e = hipGetSymbolAddress (reinterpret_cast<void **> (&ptr), HIP_SYMBOL (scratch));
  if (e != hipSuccess || !ptr) return set_error (MODES_E_HIP, "hipGetSymbolAddress(scratch): %s", hipGetErrorString (e));
  uint16_t lut[256];
  build_uc8_byte_lut (lut);
  e = hipMemcpy (ptr + kOffLut, lut, sizeof lut, hipMemcpyHostToDevice);
  if (e != hipSuccess) return set_error (MODES_E_HIP, "hipMemcpy(lut): %s", hipGetErrorString (e));
  CrcTables t;
  float uc8f[256];
  build_uc8_float_lut (uc8f);
  e = upload_constants (t.single_bit (), uc8f);            // the kernels' __constant__ tables on this device
  if (e != hipSuccess) return set_error (MODES_E_HIP, "upload_constants: %s", hipGetErrorString (e));
  g_ptr[dev] = ptr;
  return MODES_OK;
}

}  // namespace

extern "C" __attribute__ ((visibility ("default")))
int modes_convert_device_noalloc (int conv, const void *d_iq, uint16_t *d_mag, size_t nsamples, float DC_A, float DC_B,
                                  float z1_I, float z1_Q, modes_convert_result *d_result, void *stream_v)
{
  hipStream_t stream = static_cast<hipStream_t> (stream_v);
  if (conv < 0 || conv > 5) return set_error (MODES_E_ARG, "converter id %d", conv);
  if (nsamples > kMaxN)
    return set_error (MODES_E_ARG, "modes_convert_device_noalloc: %zu samples exceed the fixed scratch of %zu "
                                   "(MODES_NOALLOC_MAX_SAMPLES); use modes_convert_device()", nsamples, kMaxN);
  if ((!d_iq || !d_mag) && nsamples) return set_error (MODES_E_ARG, "null argument");
  int dev = 0;
  hipError_t e = hipGetDevice (&dev);
  if (e != hipSuccess) return set_error (MODES_E_HIP, "hipGetDevice: %s", hipGetErrorString (e));
  if (dev < 0 || dev >= kMaxDevices) return set_error (MODES_E_ARG, "device %d", dev);
  unsigned char *ptr;
  {
    std::lock_guard<std::mutex> lk (g_mu);
    const int rc = setup_device (dev, stream);
    if (rc != MODES_OK) return rc;
    ptr = g_ptr[dev];
  }
  const bool is_float = conv == kConvUc8Generic || conv == kConvSc16Generic || conv == kConvSc16Q11Generic;
  const bool dc = is_float && !(DC_A == 0.0f && DC_B == 1.0f);
  if (is_float && !dc && (z1_I != 0.0f || z1_Q != 0.0f))
    return set_error (MODES_E_ARG, "DC block off with a non-zero filter state: convert_init() never produces that");
  // from here on: `ptr`, never the array's name
  const uint16_t *lut = reinterpret_cast<const uint16_t *> (ptr + kOffLut);
  unsigned long long *power_u64 = reinterpret_cast<unsigned long long *> (ptr + kOffPower);
  float *small = reinterpret_cast<float *> (ptr + kOffSmall);
  float *f0 = reinterpret_cast<float *> (ptr + kOffFloats), *f1 = f0 + kMaxN, *f2 = f1 + kMaxN;
#define TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return set_error (MODES_E_HIP, "%s: %s", #expr, hipGetErrorString (e_)); } while (0)
  TRY (hipMemsetAsync (power_u64, 0, 32, stream));            // the accumulator and the small results start at zero
  if (nsamples) {
    if (!is_float)
      TRY (launch_convert (conv, static_cast<const uint8_t *> (d_iq), d_mag, nsamples, lut, power_u64, stream));
    else
      TRY (launch_convert_float (conv, static_cast<const uint8_t *> (d_iq), d_mag, nsamples, DC_A, DC_B, z1_I, z1_Q, f0, f1, f2,
                                 small, d_result ? small + 2 : nullptr, stream));
  }
  if (d_result) {
    hipLaunchKernelGGL (noalloc_result_kernel, dim3 (1), dim3 (64), 0, stream, conv, int (dc && nsamples), z1_I, z1_Q, power_u64, small, d_result);
    TRY (hipGetLastError ());
  }
#undef TRY
  return MODES_OK;
}
