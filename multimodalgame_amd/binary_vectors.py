"""`-binary_only`: dump every message, probability, prediction and stop bit of a deterministic dev pass
(reference: binary_vectors.py:12-135).  Same two compound datasets, `Communication` (one record per agent
message) and `Predictions` (one record per receiver step), with the reference's field names and index
convention (sender message of step i -> Index 2i, receiver message -> 2i + 1)."""
import numpy as np
import torch

from . import hdf5io
from .game import exchange
from .misc import load_hdf5


def record_types(sender_out_dim, n_classes):
    comm = np.dtype([("ExampleId", "S50"), ("AgentId", "S1"), ("Index", "<i4"), ("Target", "<i4"), ("Rank", "<i4"),
                     ("BinaryProb", "<f4", (sender_out_dim,)), ("BinaryVec", "<f4", (sender_out_dim,))])
    preds = np.dtype([("ExampleId", "S50"), ("AgentId", "S1"), ("Index", "<i4"), ("Target", "<i4"), ("Rank", "<i4"),
                      ("Predictions", "<f4", (n_classes,)), ("StopProb", "<f4", (1,)), ("StopVec", "<f4", (1,)),
                      ("StopMask", "<f4", (1,))])
    return comm, preds


def extract_binary(FLAGS, dev_file, batch_size, epoch, shuffle, sender, receiver, desc, map_labels, device):
    comm_t, preds_t = record_types(FLAGS.sender_out_dim, desc.size(0))
    comm, preds = [], []
    for batch in load_hdf5(dev_file, batch_size, epoch, shuffle, truncate_final_batch=True, map_labels=map_labels,
                           feats=(FLAGS.img_feat,), device=device):
        target, data, example_ids = batch["target"], batch[FLAGS.img_feat], batch["example_ids"]
        n = target.size(0)
        args = dict(data=data, target=target, desc=desc, train=False, break_early=not FLAGS.fixed_exchange)
        s, sen_w, rec_w, y, _, _ = exchange(sender, receiver, None, None, args)
        s_masks, s_feats, s_probs = s
        np_target = target.cpu().numpy()
        assert len(set(np_target.tolist())) == 1, "Rank only works if there is one target"     # binary_vectors.py:95-97
        single_target = int(np_target[0])
        for i, (z, pz, w, pw, yy, sf, sp, sm) in enumerate(zip(sen_w[0], sen_w[1], rec_w[0], rec_w[1], y, s_feats, s_probs, s_masks)):
            np_preds = yy.cpu().numpy()
            rank = np.abs(np_preds.argsort(1) - np_preds.shape[1])[:, single_target]             # binary_vectors.py:98
            for agent, idx, probs, vec in ((b"S", 2 * i, pz, z), (b"R", 2 * i + 1, pw, w)):
                rec = np.zeros(n, comm_t)
                rec["ExampleId"], rec["AgentId"], rec["Index"], rec["Target"], rec["Rank"] = example_ids, agent, idx, np_target, rank
                rec["BinaryProb"], rec["BinaryVec"] = probs.cpu().numpy(), vec.cpu().numpy()
                comm.append(rec)
            rec = np.zeros(n, preds_t)
            rec["ExampleId"], rec["AgentId"], rec["Index"], rec["Target"], rec["Rank"] = example_ids, b"R", 2 * i + 1, np_target, rank
            rec["Predictions"] = np_preds
            rec["StopProb"], rec["StopVec"] = sp.cpu().numpy(), sf.cpu().numpy()
            rec["StopMask"] = sm.float().cpu().numpy()
            preds.append(rec)
    with hdf5io.File(FLAGS.binary_output, "w") as f:
        f.write_struct("Communication", np.concatenate(comm) if comm else np.zeros(0, comm_t))
        f.write_struct("Predictions", np.concatenate(preds) if preds else np.zeros(0, preds_t))
    return len(comm), len(preds)
