// vss_exchange.hip — the exchange step of a sharded probe behind the C ABI (include/vssgpu.h: vss_exchange_*).
//
// north_star: "the index sharded across the 8 GPUs of one node (row-range partitions, RCCL all-gather of per-shard top-k over
// xGMI)".  The reference has no counterpart (usearch is a single-process, single-index library); what this replaces
// is the peer-copy gather of host/sharded_index.hpp whenever the shards of one host process sit on DISTINCT devices, and it is
// the same collective `duckdb-vss_amd/sharded.py` issues through torch.distributed in the one-process-per-GPU flavour.
//
// RCCL is loaded at run time (dlopen): libvssgpu.so keeps no link-time dependency on it (tests/test_cabi.py checks the
// library's needed-list), a single-GPU deployment never touches it, and a host that already carries an RCCL (PyTorch does)
// shares that copy.  Nothing here computes: one ncclAllGather of the packed per-shard blocks (row ids, then distances — the
// layout vss_packed_block_bytes describes and the search kernels fill directly), the merge is vss_merge_topk_packed_device.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <string.h>

#include <mutex>
#include <string>

#include "vssgpu.h"

namespace {

struct UniqueId { // = ncclUniqueId (rccl.h:40-43), passed BY VALUE to ncclCommInitRank
	char internal[128];
};
typedef void *Comm; // ncclComm_t

struct Rccl {
	void *handle = nullptr;
	int (*GetUniqueId)(UniqueId *) = nullptr;
	int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
	int (*CommInitAll)(Comm *, int, const int *) = nullptr;
	int (*CommDestroy)(Comm) = nullptr;
	int (*AllGather)(const void *, void *, size_t, int /*ncclDataType_t*/, Comm, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	std::string why; // why it could not be loaded
};

thread_local std::string tls_exchange_error;

Rccl &rccl() {
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		// a copy the process already maps (a PyTorch host) is found by its soname first
		for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
			r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (r.handle)
				break;
		}
		if (!r.handle) {
			const char *e = dlerror();
			r.why = std::string("librccl could not be loaded: ") + (e ? e : "not found");
			return;
		}
		bool ok = true;
		auto sym = [&](const char *name) {
			void *p = dlsym(r.handle, name);
			if (!p) {
				ok = false;
				r.why = std::string("librccl lacks ") + name;
			}
			return p;
		};
		r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
		r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
		r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
		r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
		r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
		r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
		r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
		r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
		if (!ok) {
			dlclose(r.handle);
			r.handle = nullptr;
		}
	});
	return r;
}

int fail(const std::string &what) {
	tls_exchange_error = what;
	return VSS_ERROR;
}
int check(Rccl &r, int rc, const char *what) {
	if (rc == 0 /* ncclSuccess */)
		return VSS_OK;
	return fail(std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
}
constexpr int NCCL_UINT8 = 1; // ncclUint8 (rccl.h:460)

} // namespace

struct vss_comm {
	Comm comm = nullptr;
	int n_ranks = 0, rank = 0, device = -1;
	bool owned = true; // created here (destroyed by vss_exchange_destroy) or adopted from the caller
};

extern "C" {

int vss_exchange_available(void) {
	Rccl &r = rccl();
	if (!r.handle)
		tls_exchange_error = r.why;
	return r.handle ? 1 : 0;
}

const char *vss_exchange_last_error(void) {
	return tls_exchange_error.c_str();
}

int vss_exchange_unique_id(void *id128) {
	Rccl &r = rccl();
	if (!r.handle)
		return fail(r.why);
	if (!id128)
		return fail("vss_exchange_unique_id: null buffer");
	return check(r, r.GetUniqueId(static_cast<UniqueId *>(id128)), "ncclGetUniqueId");
}

int vss_exchange_init_rank(vss_comm **out, int n_ranks, const void *id128, int rank, int device) {
	Rccl &r = rccl();
	if (!r.handle)
		return fail(r.why);
	if (!out || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks)
		return fail("vss_exchange_init_rank: bad arguments");
	if (hipSetDevice(device) != hipSuccess)
		return fail("vss_exchange_init_rank: no HIP device " + std::to_string(device));
	UniqueId id;
	memcpy(&id, id128, sizeof id);
	vss_comm *c = new vss_comm;
	c->n_ranks = n_ranks, c->rank = rank, c->device = device;
	if (check(r, r.CommInitRank(&c->comm, n_ranks, id, rank), "ncclCommInitRank") != VSS_OK) {
		delete c;
		return VSS_ERROR;
	}
	*out = c;
	return VSS_OK;
}

int vss_exchange_init_all(vss_comm **out, int n, const int *devices) {
	Rccl &r = rccl();
	if (!r.handle)
		return fail(r.why);
	if (!out || !devices || n < 1 || n > 64)
		return fail("vss_exchange_init_all: 1..64 devices");
	for (int i = 0; i != n; ++i)
		for (int j = 0; j != i; ++j)
			if (devices[i] == devices[j]) // (RCCL refuses two ranks on one device; the caller falls back to peer copies)
				return fail("vss_exchange_init_all: device " + std::to_string(devices[i]) + " named twice — one rank per device");
	Comm comms[64];
	if (check(r, r.CommInitAll(comms, n, devices), "ncclCommInitAll") != VSS_OK)
		return VSS_ERROR;
	for (int i = 0; i != n; ++i) {
		out[i] = new vss_comm;
		out[i]->comm = comms[i], out[i]->n_ranks = n, out[i]->rank = i, out[i]->device = devices[i];
	}
	return VSS_OK;
}

int vss_exchange_adopt(vss_comm **out, void *nccl_comm, int n_ranks, int rank) {
	Rccl &r = rccl();
	if (!r.handle)
		return fail(r.why);
	if (!out || !nccl_comm || n_ranks < 1 || rank < 0 || rank >= n_ranks)
		return fail("vss_exchange_adopt: bad arguments");
	vss_comm *c = new vss_comm;
	c->comm = nccl_comm, c->n_ranks = n_ranks, c->rank = rank, c->owned = false;
	*out = c;
	return VSS_OK;
}

int vss_exchange_ranks(vss_comm *c) {
	return c ? c->n_ranks : 0;
}

int vss_exchange_allgather(vss_comm *c, const void *local_block, void *gathered, uint64_t block_bytes, void *stream) {
	Rccl &r = rccl();
	if (!r.handle)
		return fail(r.why);
	if (!c || !local_block || !gathered || !block_bytes)
		return fail("vss_exchange_allgather: bad arguments");
	if (c->device >= 0 && hipSetDevice(c->device) != hipSuccess)
		return fail("vss_exchange_allgather: no HIP device " + std::to_string(c->device));
	return check(r, r.AllGather(local_block, gathered, (size_t)block_bytes, NCCL_UINT8, c->comm, (hipStream_t)stream), "ncclAllGather");
}

int vss_exchange_group_begin(void) {
	Rccl &r = rccl();
	if (!r.handle)
		return fail(r.why);
	return check(r, r.GroupStart(), "ncclGroupStart");
}

int vss_exchange_group_end(void) {
	Rccl &r = rccl();
	if (!r.handle)
		return fail(r.why);
	return check(r, r.GroupEnd(), "ncclGroupEnd");
}

int vss_exchange_destroy(vss_comm *c) {
	if (!c)
		return VSS_OK;
	int rc = VSS_OK;
	if (c->owned && c->comm) {
		Rccl &r = rccl();
		if (r.handle)
			rc = check(r, r.CommDestroy(c->comm), "ncclCommDestroy");
	}
	delete c;
	return rc;
}

} // extern "C"
