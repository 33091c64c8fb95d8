"""Data-parallel training step: one process per GPU, minibatch sharded along the batch axis,
parameters / optimizer state replicated.

Two collectives per optimizer step (RCCL over xGMI through torch.distributed backend "nccl"):
  1. a ~1.4 KB float64 all-reduce of the batch statistics the REINFORCE losses couple the shards
     through (per stream and step: n_t, sum w, sum w^2, ... -- SURVEY.md §8e option A), so every
     rank scales its gradient seeds with the statistics of the WHOLE minibatch.  Continuous messages (-nouse_binary) have
     no such coupling (loss = NLL mean over the global batch, model.py:1297-1305): this collective and the
     statistics launch are skipped, the two logged sums (rewards, hits) ride in the tail quad of the gradient buffer;
  2. one all-reduce of the flat gradient buffer of all four agents.
Gradient clipping uses the norm of the reduced gradient, so all ranks take the identical update
(model.py:1310 semantics on the global batch)."""
import os

import torch
import torch.distributed as dist


class DataParallel(object):
    """`engine` needs: forward(...), loss_stats(), backward(...), clip_step(), .stats (1-D f64 tensor),
    .flat_grads (1-D f32 tensor), .use_binary.  multimodalgame_amd.engine.Engine satisfies this on a GPU."""

    def __init__(self, engine, group=None, direct=None):
        """direct: enqueue the collectives on the engine's stream through multimodalgame_amd.rccl (default: when the
        group's backend is "nccl" and MMG_DP_DIRECT_RCCL != "0"); otherwise torch.distributed.all_reduce."""
        self.engine = engine
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.comm = None
        if direct is None:
            direct = (self.world > 1 and dist.get_backend(group) == "nccl"
                      and os.environ.get("MMG_DP_DIRECT_RCCL", "1") != "0")
        if direct:
            from . import rccl
            self.comm = rccl.try_create(engine.device, group)
        # the whole step in ONE C call when the collectives are RCCL's on the engine's stream (include/mmg.h: mmg_dp_train_step)
        self.in_library = False
        if self.comm is not None and hasattr(engine, "set_allreduce"):
            engine.set_allreduce(self.comm.all_reduce_address(), self.comm.comm)
            self.in_library = True

    def _all_reduce(self, tensor):
        if self.comm is not None:
            self.comm.all_reduce(tensor)
        else:
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)

    def train_step(self, x, target, desc, u_z=None, u_s=None, u_w=None, seed=0, full_tape=False):
        """full_tape: every sample runs all steps and the whole tape is stored (the minibatches whose log block reads it,
        model.py:1342-1542); same update."""
        e = self.engine
        if self.in_library:
            e.dp_train_step(x, target, desc, u_z, u_s, u_w, seed=seed, full_tape=full_tape, reduce=self.world > 1)
            return
        e.forward(x, target, desc, u_z, u_s, u_w, seed=seed, train=True, run_all=bool(full_tape), minimal=not full_tape, log_tape=True)
        if e.use_binary:
            e.loss_stats()
            if self.world > 1:
                self._all_reduce(e.stats)
        e.backward(x, target, desc)
        if self.world > 1:
            self._all_reduce(e.flat_grads)
        e.clip_step()


    def train_steps(self, x, target, desc, n, seed=0):
        """n consecutive minibatches (this rank's rows of each, batch-ordered) -- one C call when the collectives run in the library."""
        e = self.engine
        if self.in_library:
            e.dp_train_steps(x, target, desc, n, seed=seed, reduce=self.world > 1)
            return
        B = x.size(0) // n
        for i in range(n):
            self.train_step(x[i * B:(i + 1) * B], target[i * B:(i + 1) * B], desc, seed=seed)


def shard_range(global_batch, rank, world):
    """Contiguous B/N rows per rank."""
    assert global_batch % world == 0, "global batch must divide evenly over the ranks"
    per = global_batch // world
    return rank * per, per
