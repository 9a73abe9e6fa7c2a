/**
 * engine_exchange.hip — multi-GPU set-up: exchange buffers, P2P mailbox, RCCL communicator.
 * Part of the implementation of include/mppi_amd.h; see engine_internal.hpp for how the engine is divided and
 * engine_core.hip for the references its logic follows.
 */
#include "engine_internal.hpp"

/* ---------------------------------------------------------------- multi-GPU -------------------------------------- */
mppi_status mppi_get_exchange_buffers(mppi_handle h, void** send, void** recv, size_t* floats_per_rank)
{
  CHECK_HANDLE(h);
  if (send)
    *send = h->send_d;
  if (recv)
    *recv = h->recv_d;
  if (floats_per_rank)
    *floats_per_rank = (size_t)h->D * h->PS;
  return MPPI_OK;
}
mppi_status mppi_read_send_record(mppi_handle h, float* out)
{
  CHECK_HANDLE(h);
  if (!out)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(out, h->send_d, sizeof(float) * h->D * h->PS, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}
mppi_status mppi_write_recv_records(mppi_handle h, const float* in)
{
  CHECK_HANDLE(h);
  if (!in)
    return fail(h, MPPI_ERR_INVALID_ARG, "null");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipMemcpyAsync(h->recv_d, in, sizeof(float) * h->cfg.world_size * h->D * h->PS, hipMemcpyHostToDevice,
                            h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // `in` is the caller's
  return MPPI_OK;
}
mppi_status mppi_iteration_local(mppi_handle h)
{
  CHECK_HANDLE(h);
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  // the caller drives the optimisation loop: its iteration index (std_dev_decay) restarts with mppi_upload_state
  MPPI_TRY(iterationLocal(h, h->external_iteration++, h->last_stride));
  return flushMerge(h);  // (a caller-driven loop sees every iteration's mean: no streamed merge across its calls)
}
mppi_status mppi_iteration_merge(mppi_handle h)
{
  CHECK_HANDLE(h);
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  return iterationMerge(h);
}

static void* loadRccl(std::string& err)
{
  static void* lib = nullptr;
  if (lib)
    return lib;
  // The communicator must live on the SAME HIP runtime as this library's streams and buffers.  A host application may
  // carry a second ROCm stack (PyTorch wheels bundle their own libamdhip64 / librccl), and a plain dlopen("librccl.so")
  // would hand back that copy.  So: first the librccl that sits next to the libamdhip64 this library is linked to (by
  // absolute path), then the usual names.
  std::vector<std::string> candidates;
  Dl_info info{};
  if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname)
  {
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos)
    {
      dir.resize(slash);
      candidates.push_back(dir + "/librccl.so.1");
      candidates.push_back(dir + "/librccl.so");
    }
  }
  candidates.push_back("/opt/rocm/lib/librccl.so.1");
  candidates.push_back("/opt/rocm/lib/librccl.so");
  candidates.push_back("librccl.so.1");
  candidates.push_back("librccl.so");
  for (const std::string& n : candidates)
  {
    lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (lib)
      return lib;
  }
  err = std::string("cannot dlopen librccl: ") + dlerror();
  return nullptr;
}

/* ---------------------------------------------------------------- P2P mailbox exchange ---------------------------- */
static mppi_status ensureMailbox(mppi_handle h)
{
  if (h->mbox_d)
    return MPPI_OK;
  const int world = h->cfg.world_size;
  if (world > 16)
    return fail(h, MPPI_ERR_UNSUPPORTED, "P2P mailbox exchange supports up to 16 ranks");
  const size_t dps = (size_t)h->D * h->PS;
  // records | flags [2][world], ticket counter (+ 3 words of padding) | aux arrays [2][MAILBOX_AUX_FLOATS] | aux flags [2][world]
  h->mbox_aux_off = 2 * world * dps + (size_t)(2 * world + 4);  // in 4-byte words from the base
  h->mbox_aux_off = (h->mbox_aux_off + 3) & ~(size_t)3;
  h->mbox_bytes = sizeof(float) * (h->mbox_aux_off + 2 * (size_t)kernels::MAILBOX_AUX_FLOATS + 2 * (size_t)world);
  h->mbox_bytes = (h->mbox_bytes + 4095) & ~(size_t)4095;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  // uncached device memory where the runtime offers it (the mailbox is written by other agents); every access to it is a
  // system-scope atomic anyway, so ordinary device memory is a correct fallback
  hipError_t e = hipExtMallocWithFlags((void**)&h->mbox_d, h->mbox_bytes, hipDeviceMallocUncached);
  h->mbox_uncached = (e == hipSuccess);
  if (e != hipSuccess)
  {
    (void)hipGetLastError();
    HIP_TRY(h, hipMalloc((void**)&h->mbox_d, h->mbox_bytes));
  }
  HIP_TRY(h, hipMemsetAsync(h->mbox_d, 0, h->mbox_bytes, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}

/**
 * A new exchange session starts at sequence number 1 again, and the merge kernel waits for flag == sequence number: flags
 * and records left by an earlier session (one that ended after 1-3 iterations would match sequence 1 / 2 of the new one)
 * are cleared here.  Called where a session begins BEFORE a peer of the new session can reach the mailbox: when its IPC
 * handle is exported (peers map it after that), and by mppi_p2p_connect_local (in-process ranks connect before their first
 * exchange).
 */
static mppi_status resetMailboxSession(mppi_handle h)
{
  if (!h->mbox_d || (h->xseq == 0 && h->aseq == 0))
    return MPPI_OK;  // fresh (zeroed at allocation) or never used since the last reset
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipMemsetAsync(h->mbox_d, 0, h->mbox_bytes, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->xseq = 0;
  h->aseq = 0;
  return MPPI_OK;
}

mppi_status mppi_p2p_mailbox_handle(mppi_handle h, void* out_bytes, size_t capacity, size_t* nbytes)
{
  CHECK_HANDLE(h);
  if (!out_bytes || capacity < sizeof(hipIpcMemHandle_t))
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_p2p_mailbox_handle: buffer too small (needs 64 bytes)");
  MPPI_TRY(ensureMailbox(h));
  // Exporting the handle has no side effect on a LIVE session (a caller that asks twice, a peer that maps late): the mailbox
  // is only cleared when no session is connected — a new one starts with mppi_p2p_reset (or on a handle that never ran)
  if (!h->p2p_ready)
    MPPI_TRY(resetMailboxSession(h));
  hipIpcMemHandle_t ipc;
  hipError_t e = hipIpcGetMemHandle(&ipc, h->mbox_d);
  if (e != hipSuccess && h->mbox_uncached)
  {  // this runtime does not export uncached allocations: fall back to ordinary device memory
    (void)hipGetLastError();
    (void)hipFree(h->mbox_d);
    h->mbox_d = nullptr;
    HIP_TRY(h, hipMalloc((void**)&h->mbox_d, h->mbox_bytes));
    h->mbox_uncached = false;
    HIP_TRY(h, hipMemsetAsync(h->mbox_d, 0, h->mbox_bytes, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    e = hipIpcGetMemHandle(&ipc, h->mbox_d);
  }
  if (e != hipSuccess)
    return fail(h, MPPI_ERR_HIP, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e) +
                                     " (multi-process GPU sharing needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this driver)");
  memcpy(out_bytes, &ipc, sizeof(ipc));
  if (nbytes)
    *nbytes = sizeof(ipc);
  return MPPI_OK;
}

/** the exchange-failure mark stats_d[z][6] is sticky on the device (no kernel clears it, reduce_kernels.hpp: combineWave): a new
 *  session starts without it */
static mppi_status clearExchangeFailure(mppi_handle h)
{
  h->exchange_failed = false;
  if (!h->stats_d)
    return MPPI_OK;
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  for (int z = 0; z < h->D; z++)
    HIP_TRY(h, hipMemsetAsync(h->stats_d + (size_t)z * kernels::STATS_STRIDE + 6, 0, sizeof(float), h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return MPPI_OK;
}

mppi_status mppi_p2p_reset(mppi_handle h)
{
  CHECK_HANDLE(h);
  h->p2p_ready = false;
  MPPI_TRY(clearExchangeFailure(h));
  return resetMailboxSession(h);
}

mppi_status mppi_p2p_connect(mppi_handle h, const void* handles, size_t stride_bytes)
{
  CHECK_HANDLE(h);
  const int world = h->cfg.world_size;
  if (!handles || stride_bytes < sizeof(hipIpcMemHandle_t))
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_p2p_connect: handles[world] with a stride of at least 64 bytes expected");
  MPPI_TRY(ensureMailbox(h));
  /* A session's sequence numbers start at 1 and the merge waits for flag == sequence number, so a mailbox that still holds the
   * flags and records of an earlier session would let this one pass its waits early (stale or half-written peer records merged
   * silently).  The mailbox cannot be cleared HERE — a peer of the new session that connected first may already have posted
   * into it — only before its handle is exported, which mppi_p2p_mailbox_handle does on a handle without a live session.  A
   * live or used session therefore has to be ended explicitly first: mppi_p2p_reset, then export, then connect. */
  if (h->p2p_ready || h->xseq != 0 || h->aseq != 0)
    return fail(h, MPPI_ERR_STATE, "mppi_p2p_connect: this handle has a live (or used) exchange session; call mppi_p2p_reset on "
                                   "every rank, export the mailbox handles again (mppi_p2p_mailbox_handle) and then connect");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  for (int p = 0; p < world; p++)
  {
    if (p == h->cfg.rank)
    {
      h->peer_mbox[p] = h->mbox_d;
      continue;
    }
    if (h->peer_opened[p] && h->peer_mbox[p])
    {  // a reconnect: the mapping of the previous session goes first
      (void)hipIpcCloseMemHandle(h->peer_mbox[p]);
      h->peer_opened[p] = false;
      h->peer_mbox[p] = nullptr;
    }
    hipIpcMemHandle_t ipc;
    memcpy(&ipc, (const char*)handles + (size_t)p * stride_bytes, sizeof(ipc));
    void* ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, ipc, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess)
      return fail(h, MPPI_ERR_COMM, "hipIpcOpenMemHandle for rank " + std::to_string(p) + ": " + hipGetErrorString(e));
    h->peer_mbox[p] = (float*)ptr;
    h->peer_opened[p] = true;
  }
  h->xseq = 0;
  h->aseq = 0;
  MPPI_TRY(clearExchangeFailure(h));
  h->p2p_ready = true;
  return MPPI_OK;
}

mppi_status mppi_p2p_connect_local(mppi_handle h, const mppi_handle* peers)
{
  CHECK_HANDLE(h);
  const int world = h->cfg.world_size;
  if (!peers)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_p2p_connect_local: null");
  MPPI_TRY(ensureMailbox(h));
  MPPI_TRY(resetMailboxSession(h));
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  for (int p = 0; p < world; p++)
  {
    mppi_handle q = peers[p];
    if (!q || q->cfg.world_size != world || q->cfg.rank != p || q->D != h->D || q->PS != h->PS)
      return fail(h, MPPI_ERR_INVALID_ARG, "mppi_p2p_connect_local: peers[p] must be the handle of rank p of the same problem");
    if (q != h)
    {
      std::lock_guard<std::recursive_mutex> peer_lock(q->mu);
      MPPI_TRY(ensureMailbox(q) == MPPI_OK ? MPPI_OK : fail(h, MPPI_ERR_HIP, "peer mailbox allocation failed"));
      HIP_TRY(h, hipSetDevice(h->cfg.device));
      if (q->cfg.device != h->cfg.device)
      {
        int can = 0;
        HIP_TRY(h, hipDeviceCanAccessPeer(&can, h->cfg.device, q->cfg.device));
        if (!can)
          return fail(h, MPPI_ERR_COMM, "no peer access between device " + std::to_string(h->cfg.device) + " and " +
                                            std::to_string(q->cfg.device));
        const hipError_t e = hipDeviceEnablePeerAccess(q->cfg.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
          return fail(h, MPPI_ERR_COMM, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        (void)hipGetLastError();
      }
    }
    h->peer_mbox[p] = q->mbox_d;
  }
  h->xseq = 0;
  h->aseq = 0;
  MPPI_TRY(clearExchangeFailure(h));
  h->p2p_ready = true;
  return MPPI_OK;
}

mppi_status mppi_rccl_unique_id(void* out_bytes, size_t capacity, size_t* nbytes)
{
  if (!out_bytes || capacity < 128)
    return MPPI_ERR_INVALID_ARG;
  std::string err;
  void* lib = loadRccl(err);
  if (!lib)
    return fail(nullptr, MPPI_ERR_COMM, err);
  typedef int (*fn_t)(void*);
  fn_t f = (fn_t)dlsym(lib, "ncclGetUniqueId");
  if (!f)
    return fail(nullptr, MPPI_ERR_COMM, "ncclGetUniqueId not found");
  const int rc = f(out_bytes);  // ncclUniqueId is 128 bytes
  if (nbytes)
    *nbytes = 128;
  return rc == 0 ? MPPI_OK : fail(nullptr, MPPI_ERR_COMM, "ncclGetUniqueId failed");
}

mppi_status mppi_comm_init_rccl(mppi_handle h, const void* unique_id, size_t nbytes)
{
  CHECK_HANDLE(h);
  if (!unique_id || nbytes != 128)
    return fail(h, MPPI_ERR_INVALID_ARG, "mppi_comm_init_rccl: unique id must be 128 bytes");
  std::string err;
  void* lib = loadRccl(err);
  if (!lib)
    return fail(h, MPPI_ERR_COMM, err);
  struct Id
  {
    char b[128];
  } id;
  memcpy(id.b, unique_id, 128);
  typedef int (*init_fn)(void**, int, Id, int);
  init_fn f = (init_fn)dlsym(lib, "ncclCommInitRank");
  g_ncclAllGather = (nccl_allgather_fn)dlsym(lib, "ncclAllGather");
  if (!f || !g_ncclAllGather)
    return fail(h, MPPI_ERR_COMM, "ncclCommInitRank / ncclAllGather not found in librccl");
  HIP_TRY(h, hipSetDevice(h->cfg.device));
  const int rc = f(&h->comm, h->cfg.world_size, id, h->cfg.rank);
  if (rc != 0)
    return fail(h, MPPI_ERR_COMM, "ncclCommInitRank failed with code " + std::to_string(rc));
  h->rccl_lib = lib;
  return MPPI_OK;
}
