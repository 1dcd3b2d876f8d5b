// C-ABI: caller-owned host buffers (the reference keeps its particles in a host std::vector<PFSample> that is
// allocated once, particle_filter.cpp:62-89, and hands it to the sensor model every cycle).
// ---------------------------------------------------------------------- host buffers
// A pageable buffer crosses PCIe through the ENGINE's bounce buffer (h2d_from_host, host_common.inl: the runtime's own
// on-the-fly pinning keeps stale pins of buffers that were freed and allocated again); handed to the runtime directly
// (BPF_OPT_HOST_DIRECT_PAGEABLE) 3.2 MB go up at the link's rate too
// (64 us, tools/ubench/pcie_probe.hip), but the calling thread is busy inside the call for that long and a download
// into it takes 44 us per 0.8 MB instead of 22; a buffer pinned with hipHostRegister is read and written by the copy
// engine -- or by a kernel -- directly, with the calling thread free.  Registration costs 60 us to a millisecond, so
// it is done ONCE per buffer and kept.  It is the OWNER's statement that
// the memory stays allocated until bpf_host_buffer_unregister / bpf_destroy: the driver follows a registered range
// by virtual address, and a range that has been freed (or freed and re-allocated) under a live registration makes
// the next DMA fault.  That is why the engine does not cache registrations by pointer on its own
// (BPF_OPT_HOST_AUTO_REGISTER is the caller's explicit promise for every buffer it hands over).
namespace
{
bpf_engine::HostReg* host_reg_find(bpf_engine* e, const void* ptr, size_t bytes)
{
  const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
  for (auto& r : e->host_regs)
    if (a >= r.base && a + bytes <= r.base + r.bytes)
      return &r;
  return nullptr;
}

int host_reg_add(bpf_engine* e, void* ptr, size_t bytes, bool automatic)
{
  const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
  // a range that overlaps an older registration without lying inside it: the buffer was re-allocated or grew
  for (size_t i = 0; i < e->host_regs.size();)
  {
    auto& r = e->host_regs[i];
    if (a < r.base + r.bytes && r.base < a + bytes)
    {
      if (!r.automatic)  // an owner's registration is never dropped behind the owner's back
        return e->fail(BPF_ERR_INVALID_ARGUMENT, "host buffer overlaps a registered range");
      (void)hipHostUnregister(reinterpret_cast<void*>(r.base));
      e->host_regs.erase(e->host_regs.begin() + (long)i);
    }
    else
      ++i;
  }
  if (hipHostRegister(ptr, bytes, hipHostRegisterDefault) != hipSuccess)
  {
    (void)hipGetLastError();
    return e->fail(BPF_ERR_HIP, "hipHostRegister refused the buffer");
  }
  void* dv = nullptr;
  if (hipHostGetDevicePointer(&dv, ptr, 0) != hipSuccess)
  {
    (void)hipGetLastError();
    dv = nullptr;
  }
  e->host_regs.push_back(bpf_engine::HostReg{ a, bytes, automatic, reinterpret_cast<uintptr_t>(dv) });
  return BPF_OK;
}

// true when [ptr, ptr + bytes) may be handed to the copy engine as pinned memory
bool host_buffer_pinned(bpf_engine* e, void* ptr, size_t bytes)
{
  if (host_reg_find(e, ptr, bytes))
    return true;
  if (!e->host_auto_register || bytes < (size_t)64 * 1024)
    return false;
  return host_reg_add(e, ptr, bytes, true) == BPF_OK;
}

void host_buffers_release(bpf_engine* e)
{
  for (auto& r : e->host_regs)
    (void)hipHostUnregister(reinterpret_cast<void*>(r.base));
  e->host_regs.clear();
  for (auto ev : e->seam_ev)
    (void)hipEventDestroy(ev);
  e->seam_ev.clear();
  if (e->copy_up)
    (void)hipStreamDestroy(e->copy_up);
  e->copy_up = nullptr;
  for (int h = 0; h < 2; ++h)
  {
    if (e->bounce_ev[h])
      (void)hipEventDestroy(e->bounce_ev[h]);
    e->bounce_ev[h] = nullptr;
    e->bounce_busy[h] = false;
  }
}

int seam_resources(bpf_engine* e)
{
  if (e->copy_up)
    return BPF_OK;
  HIPCHK(e, hipStreamCreateWithFlags(&e->copy_up, hipStreamNonBlocking));
  e->seam_ev.resize((size_t)kSeamMaxChunks, nullptr);
  for (auto& ev : e->seam_ev)
    HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIPCHK(e, e->d_seam_partials.reserve((size_t)kSeamMaxChunks * kSeamMaxBlocks));
  HIPCHK(e, e->h_seam_flags.reserve((size_t)kSeamMaxChunks));
  HIPCHK(e, e->h_seam_totals.reserve((size_t)kSeamMaxChunks));
  for (int i = 0; i < kSeamMaxChunks; ++i)
    e->h_seam_flags.p[i] = 0ull;
  return BPF_OK;
}
}  // namespace

int bpf_host_buffer_register(bpf_engine* e, void* ptr, size_t bytes)
{
  if (!e || !ptr || bytes == 0)
    return e ? e->fail(BPF_ERR_INVALID_ARGUMENT, "host buffer: null or empty") : BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  if (bpf_engine::HostReg* r = host_reg_find(e, ptr, bytes))
  {
    r->automatic = false;  // now the owner's
    return BPF_OK;
  }
  return host_reg_add(e, ptr, bytes, false);
}

int bpf_host_buffer_unregister(bpf_engine* e, void* ptr)
{
  if (!e || !ptr)
    return BPF_ERR_INVALID_ARGUMENT;
  HIPCHK(e, hipSetDevice(e->device));
  const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
  for (size_t i = 0; i < e->host_regs.size(); ++i)
    if (e->host_regs[i].base == a)
    {
      // nothing of the engine may still be reading or writing it
      HIPCHK(e, hipStreamSynchronize(e->stream));
      if (e->copy_up)
        HIPCHK(e, hipStreamSynchronize(e->copy_up));
      HIPCHK(e, hipHostUnregister(ptr));
      e->host_regs.erase(e->host_regs.begin() + (long)i);
      return BPF_OK;
    }
  return e->fail(BPF_ERR_INVALID_ARGUMENT, "host buffer: not a registered base address");
}

int bpf_host_buffer_is_registered(bpf_engine* e, const void* ptr, size_t bytes)
{
  if (!e || !ptr)
    return 0;
  return host_reg_find(e, ptr, bytes) != nullptr ? 1 : 0;
}

int bpf_score_last_form(bpf_engine* e, int* form_out)
{
  if (!e || !form_out)
    return BPF_ERR_INVALID_ARGUMENT;
  *form_out = e->last_score_form;
  return BPF_OK;
}

int bpf_kld_last_form(bpf_engine* e, int* form_out)
{
  if (!e || !form_out)
    return BPF_ERR_INVALID_ARGUMENT;
  *form_out = e->kld_last_form;
  return BPF_OK;
}

int bpf_seam_last_plan(bpf_engine* e, int* chunks_out, int* pinned_out)
{
  if (!e)
    return BPF_ERR_INVALID_ARGUMENT;
  if (chunks_out)
    *chunks_out = e->last_seam_chunks;
  if (pinned_out)
    *pinned_out = e->last_seam_registered ? 1 : 0;
  return BPF_OK;
}
