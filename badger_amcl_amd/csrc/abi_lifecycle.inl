// C-ABI: engine lifecycle and stream binding.
int bpf_create(int device_ordinal, bpf_engine** out)
{
  if (!out)
    return BPF_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return BPF_ERR_HIP;  // no GPU: the product path fails loudly, there is no CPU fallback
  if (device_ordinal < 0 || device_ordinal >= count)
    return BPF_ERR_INVALID_ARGUMENT;
  if (hipSetDevice(device_ordinal) != hipSuccess)
    return BPF_ERR_HIP;
  bpf_engine* e = new bpf_engine();
  e->device = device_ordinal;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess)
    e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess)
  {
    delete e;
    return BPF_ERR_HIP;
  }
  e->stream = e->own_stream;
  lcg_tables(e->jump);
  *out = e;
  return BPF_OK;
}

void bpf_destroy(bpf_engine* e)
{
  if (!e)
    return;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  for (auto& s : e->ring)
  {
    s.host.release();
    s.dev.release();
    if (s.done)
      (void)hipEventDestroy(s.done);
  }
  collective_release(e);
  mailbox_release(e);
  host_buffers_release(e);
  if (e->targets_read)
    (void)hipEventDestroy(e->targets_read);
  for (auto ev : e->ev_start)
    (void)hipEventDestroy(ev);
  for (auto ev : e->ev_stop)
    (void)hipEventDestroy(ev);
  // every DevBuf / PinnedBuf member frees itself when the engine is deleted (the device is selected above)
  if (e->own_stream)
    (void)hipStreamDestroy(e->own_stream);
  delete e;
}

const char* bpf_error_string(int code)
{
  switch (code)
  {
    case BPF_OK: return "ok";
    case BPF_ERR_INVALID_ARGUMENT: return "invalid argument";
    case BPF_ERR_NOT_CONFIGURED: return "map, model or filter not configured";
    case BPF_ERR_HIP: return "HIP runtime error (or no GPU present)";
    case BPF_ERR_UNSUPPORTED: return "unsupported on the device path";
    case BPF_ERR_CDF_MISS: return "CDF search found no interval (reference asserts)";
    case BPF_ERR_LUT_LEVELS: return "distance LUT has too many distinct values";
    case BPF_ERR_BEAM_STEP: return "beam model step is zero (reference never returns)";
    case BPF_ERR_CAPACITY: return "capacity exceeded";
    case BPF_ERR_EXCHANGE: return "exchange between the shards failed (a peer did not answer in time)";
    default: return "unknown";
  }
}

const char* bpf_last_error_message(const bpf_engine* e)
{
  return e ? e->last_error.c_str() : "null engine";
}

int bpf_set_stream(bpf_engine* e, void* hip_stream)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->stream = (hip_stream == BPF_OWN_STREAM) ? e->own_stream : reinterpret_cast<hipStream_t>(hip_stream);
  return BPF_OK;
}

int bpf_synchronize(bpf_engine* e)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return BPF_OK;
}
