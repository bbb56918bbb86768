"""Input side of the training step on the device: the data set lives in HBM in the reference's own storage format --
`imgs` uint8 [N, H, W, 3] + `labels` int64 [N], exactly what reference src/utils/hdf5.py:35-97 writes and src/data_util.py:102-142 reads
back with `load_data_in_memory` -- and a training "basket" (reference src/worker.py:194-208: one loader batch of
batch_size * d_updates_per_step * acml_steps samples, torch.split into micro-batches) is ONE gather kernel that also applies the random
horizontal flip. The discriminator mirrors take the uint8 micro-batches directly (functional.u8_to_nhwc normalises on load), so no fp32
image and no host-to-device copy exists on the training path. ImageNet-128 as uint8 is 63 GB of the 288 GB."""
import numpy as np
import torch

from . import _lib as L


class DeviceDataset:
    def __init__(self, imgs, labels, device="cuda", random_flip=True, seed=0, rank=None, world_size=None):
        """rank / world_size: the data-parallel shard this process draws from (default: torch.distributed's, 0 / 1 without it). Every rank
        keeps the SAME per-epoch permutation (generator seeded with `seed`) and takes perm[rank::world] of it -- the DistributedSampler of the
        reference (src/loader.py:153-171, shuffle=True, drop_last=True); the flip stream is seeded with seed + 1 + rank."""
        imgs = torch.as_tensor(np.ascontiguousarray(imgs) if isinstance(imgs, np.ndarray) else imgs)
        if imgs.dtype != torch.uint8 or imgs.dim() != 4 or imgs.shape[3] != 3:
            raise RuntimeError("expected uint8 images [N, H, W, 3] (the HDF5 layout of the reference)")
        self.device = torch.device(device)
        self.imgs = imgs.to(self.device).contiguous()
        self.labels = torch.as_tensor(labels).long().to(self.device).contiguous()
        assert self.labels.shape[0] == self.imgs.shape[0]
        self.random_flip = random_flip
        if rank is None or world_size is None:
            import torch.distributed as dist
            ddp = dist.is_available() and dist.is_initialized()
            rank = (dist.get_rank() if ddp else 0) if rank is None else rank
            world_size = (dist.get_world_size() if ddp else 1) if world_size is None else world_size
        if not (0 <= rank < world_size):
            raise RuntimeError(f"DeviceDataset: rank {rank} outside world of {world_size}")
        self.rank, self.world_size = int(rank), int(world_size)
        self.gen = torch.Generator(device=self.device)          # the shared permutation stream: identical on every rank
        self.gen.manual_seed(seed)
        self.flip_gen = torch.Generator(device=self.device)     # per-rank augmentation stream
        self.flip_gen.manual_seed(seed + 1 + self.rank)
        self._perm, self._pos, self.epoch = None, 0, 0

    @classmethod
    def from_hdf5(cls, path, **kw):
        """reference src/data_util.py:103-111 (`load_data_in_memory`): imgs / labels data sets of the file make_hdf5 wrote."""
        try:
            import h5py
        except ImportError as e:      # not in this image; the format is plain arrays, so anything that yields them works with __init__
            raise RuntimeError("h5py is needed to read " + path + " (pass the arrays to DeviceDataset(imgs, labels) instead)") from e
        with h5py.File(path, "r") as f:
            return cls(f["imgs"][:], f["labels"][:], **kw)

    def __len__(self):
        return self.imgs.shape[0]

    def shard_len(self):
        """samples per epoch of THIS rank (DistributedSampler(drop_last=True): the ragged tail is dropped so every rank steps equally often)"""
        return len(self) // self.world_size

    def gather(self, idx, flip=None):
        """images [B, H, W, 3] uint8 and labels [B] of the given sample indices (device tensors); flip: optional uint8 mask [B]."""
        idx = idx.to(self.device).long().contiguous()
        B = idx.numel()
        N, H, W, _ = self.imgs.shape
        out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=self.device)
        lab = torch.empty(B, dtype=torch.int64, device=self.device)
        fl = None if flip is None else flip.to(self.device).to(torch.uint8).contiguous()
        L.call("sg_gather_images_u8", L.ptr(self.imgs), L.ptr(idx), L.ptr(fl), L.ptr(out), B, H, W, L.ptr(self.labels), L.ptr(lab), L.stream())
        return out, lab

    def _next_indices(self, n):
        """sequential sampling without replacement over a per-epoch permutation (DataLoader(shuffle=True, drop_last=True) semantics)."""
        N = len(self)
        per_rank = self.shard_len()
        if n > per_rank:
            raise RuntimeError(f"DeviceDataset: a basket of {n} samples exceeds this rank's shard of {per_rank} "
                               f"({N} images over {self.world_size} rank(s)): lower batch_size * d_updates_per_step * acml_steps")
        if self._perm is None or self._pos + n > per_rank:
            full = torch.randperm(N, device=self.device, generator=self.gen)          # same draw on every rank
            self._perm = full[self.rank:per_rank * self.world_size:self.world_size]   # perm[rank::world], tail dropped
            self._pos = 0
            self.epoch += 1
        idx = self._perm[self._pos:self._pos + n]
        self._pos += n
        return idx

    def sample_data_basket(self, batch_size, num_micro_batches):
        """reference src/worker.py:194-208: (tuple of image micro-batches, tuple of label micro-batches), each of batch_size samples."""
        n = batch_size * num_micro_batches
        idx = self._next_indices(n)
        flip = None
        if self.random_flip:
            flip = (torch.rand(n, device=self.device, generator=self.flip_gen) < 0.5).to(torch.uint8)      # transforms.RandomHorizontalFlip(p=0.5)
        imgs, labs = self.gather(idx, flip)
        return torch.split(imgs, batch_size), torch.split(labs, batch_size)
