"""Serving-side consumer of the forward path: ``ModelInference`` of the reference
(recommendation/inference_service.py:36-93) on the MI355X kernels.

Same constructor argument, same two calls, same results: a list of record strings
"user country l1 click1 l2 click2" goes through the record tokenizer, features(), forward() and
the softmax/top-k head; ``inference`` returns each sample's top-6 colour ids SORTED ascending and
joined by ',', ``inference_click`` the top-1 colour id, or -1 where its probability is not above
``threshold`` (recommendation/config.py:60, 0.1).  The gRPC server, the model hot-swap watcher
and the database writes around it are control plane and stay outside.
"""
import torch

from . import checkpoint
from .r_model import COLUMNS, RecommendationModel, parse_records

threshold = 0.1        # recommendation/config.py:60


class ModelInference(object):
    def __init__(self, checkpoint_dir=None, device=None, model=None):
        self.checkpoint_dir = checkpoint_dir
        self.m = model if model is not None else RecommendationModel(colour_count=128, recommend_num=6,
                                                                     user_count=100, country_count=20, device=device)
        self._loaded = False

    def _features(self, record):
        if isinstance(record, (str, bytes)):
            record = [record]
        text = '\n'.join(r.decode() if isinstance(r, bytes) else r for r in record)
        cols, _ = parse_records(text, 6, 6, len(record))
        if len(cols[0]) != len(record):
            raise ValueError("empty record in the request")
        return {k: torch.from_numpy(cols[i]) for i, k in enumerate(COLUMNS[:6])}

    def _forward(self, record):
        fs = self.m.features(self._features(record))
        with torch.no_grad():
            last = self.m.forward(fs)
        if not self._loaded:               # the variables exist after the first forward, as in a fresh TF graph
            self._loaded = True
            if self.checkpoint_dir is not None and checkpoint.latest(self.checkpoint_dir) is not None:
                checkpoint.load(self.checkpoint_dir, self.m)
                with torch.no_grad():
                    last = self.m.forward(self.m.features(self._features(record)))
        return last

    def inference(self, record):
        indices, _ = self.m.output(self._forward(record))
        return [','.join(str(x) for x in sorted(row)) for row in indices.cpu().tolist()]

    def inference_click(self, record):
        idx, val = self.m.top_one_output(self._forward(record))
        idx, val = idx.reshape(-1).cpu().tolist(), val.reshape(-1).cpu().tolist()
        return [i if v > threshold else -1 for i, v in zip(idx, val)]

    def close(self):
        pass
