"""Streaming predictions writer for the validation / virtual-screening loops.

The reference formats every batch's predictions on the host inside the loop (a device->host sync per
batch, /root/reference/point_vs/models/point_neural_network_base.py:247-325) and appends them to the
predictions file every `log_interval` batches (:492-499). Here the loop only ENQUEUES a batch: the
scores are copied to a pinned host buffer asynchronously and a writer thread waits for the copy,
formats the same lines and appends them at the same cadence - the GPU never waits for the host, and
the file on disk is byte-identical to what the synchronous loop would have written.

Line formats (point_neural_network_base.py:287-325):
    labelled            '{label:.3f} | {prediction:.3f} {receptor} {ligand}'      (label int()ed for classification)
    unlabelled          '{prediction:.3f} | {receptor} {ligand}'
    multi_regression    '{label:.3f} | {prediction:.3f} {receptor} {ligand} | {metric}'   (labelled targets > -0.5)
                        '{p0:.3f} {p1:.3f} {p2:.3f} | {receptor} {ligand}'        (unlabelled)
"""
import queue
import threading
from pathlib import Path

import numpy as np
import torch

_METRICS = ('pki', 'pkd', 'ic50')


def format_lines(task, y_pred, y_true, receptors, ligands):
    """The reference's line formats for one batch (numpy inputs; y_true None = unlabelled)."""
    n = len(receptors)
    if task == 'multi_regression':
        y_pred = np.asarray(y_pred).reshape(-1, 3)
        if y_true is None:
            return ['{0:.3f} {1:.3f} {2:.3f} | {3} {4}'.format(*y_pred[i], receptors[i], ligands[i]) for i in range(n)]
        # As the reference does (:262-270, :289-296): labelled targets (> -0.5) are flattened in
        # (sample, metric) order and the i-th of them is printed beside the i-th sample's file names,
        # n lines per batch - a quirk of the reference kept for byte-identical files.
        y_true = np.asarray(y_true).reshape(-1, 3)
        keep = np.where(y_true > -0.5)
        metrics = np.array([_METRICS] * n)[keep]
        y_pred, y_true = y_pred[keep], y_true[keep]
        return ['{0:.3f} | {1:.3f} {2} {3} | {4}'.format(float(y_true[i]), y_pred[i], receptors[i], ligands[i],
                                                        metrics[i]) for i in range(n)]
    y_pred = np.asarray(y_pred).reshape(-1)
    if y_true is None:
        return ['{0:.3f} | {1} {2}'.format(y_pred[i], receptors[i], ligands[i]) for i in range(n)]
    y_true = np.asarray(y_true).reshape(-1)
    cast = int if task == 'classification' else float
    return ['{0:.3f} | {1:.3f} {2} {3}'.format(cast(y_true[i]), y_pred[i], receptors[i], ligands[i])
            for i in range(n)]


def rank_part(path, rank):
    """Part file of one rank of a data-parallel validation run (joined by merge_rank_files)."""
    path = Path(path)
    return path.with_name(f'{path.name}.rank{rank}')


def merge_rank_files(path, world):
    """Joins the ranks' part files in rank order into `path` (ranks hold contiguous shares of the set, so
    this is the order one process would have written) and removes them. Called by rank 0 between two
    barriers; a rank with an empty share has written no part."""
    path = Path(path)
    tmp = path.with_name(path.name + '.joining')
    with open(tmp, 'wb') as out:
        for r in range(world):
            part = rank_part(path, r)
            if part.is_file():
                out.write(part.read_bytes())
                part.unlink()
    tmp.replace(path)


def read_predictions(path):
    """Rows of a labelled predictions file: (y_true, y_pred, receptor, ligand[, metric])."""
    rows = []
    for line in Path(path).read_text().splitlines():
        parts = line.split()
        if len(parts) >= 5 and parts[1] == '|':
            rows.append((float(parts[0]), float(parts[2]), parts[3], parts[4]) + tuple(parts[6:7]))
    return rows


def top_n(path, n=1):
    """Fraction of receptors with an active among their n best-scored poses: the reference's model-selection
    metric for pose models (/root/reference/point_vs/analysis/top_n.py:32-49, used by val(): poses grouped by
    receptor, sorted by predicted score, descending)."""
    by_rec = {}
    for y_true, y_pred, rec, _lig, *_ in read_predictions(path):
        by_rec.setdefault(rec, []).append((y_pred, int(y_true)))
    if not by_rec:
        return 0.0
    hits = sum(1 for poses in by_rec.values()
               if sum(label for _, label in sorted(poses, key=lambda t: t[0], reverse=True)[:n]))
    return hits / len(by_rec)


def regression_pearson(path):
    """(r, p) of predicted against true affinities: /root/reference/point_vs/utils.py:189-198."""
    from scipy.stats import pearsonr
    rows = read_predictions(path)
    return pearsonr([r[0] for r in rows], [r[1] for r in rows])


class PredictionsWriter:
    """with PredictionsWriter(path, task) as w:  w.submit(y_pred_device, y_true, receptors, ligands)"""

    def __init__(self, path, task, flush_every=10, ring=8):
        self.path, self.task, self.flush_every = Path(path).expanduser(), task, max(1, int(flush_every))
        self.path.parent.mkdir(parents=True, exist_ok=True)
        if self.path.is_file():
            self.path.unlink()
        self._q = queue.Queue()
        self._free = threading.Semaphore(ring)        # bounds the pinned buffers in flight
        self._error = None
        self.lines_written = 0
        self._thread = threading.Thread(target=self._run, name='pvs-predictions', daemon=True)
        self._thread.start()

    def _stage(self, t):
        """Device tensor -> (pinned host copy in flight, event); host data passes through."""
        if t is None or not torch.is_tensor(t):
            return t, None
        t = t.detach()
        if not t.is_cuda:
            return t, None
        host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        host.copy_(t, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(t.device))
        return host, ev

    def submit(self, y_pred, y_true, receptors, ligands):
        """Enqueues one batch; returns immediately (blocks only when `ring` batches are already
        waiting for their copies, i.e. the device is that far behind the loop)."""
        if self._error is not None:
            raise self._error
        self._free.acquire()
        pred, ev_p = self._stage(y_pred)
        true, ev_t = self._stage(y_true)
        self._q.put((pred, ev_p, true, ev_t, list(receptors), list(ligands)))

    def _run(self):
        pending, batches = [], 0
        try:
            while True:
                item = self._q.get()
                if item is None:
                    break
                pred, ev_p, true, ev_t, receptors, ligands = item
                for ev in (ev_p, ev_t):
                    if ev is not None:
                        ev.synchronize()         # only this thread waits for the copy
                pred = pred.numpy() if torch.is_tensor(pred) else np.asarray(pred)
                true = None if true is None else (true.numpy() if torch.is_tensor(true) else np.asarray(true))
                pending += format_lines(self.task, pred, true, receptors, ligands)
                self._free.release()
                batches += 1
                if batches % self.flush_every == 0:
                    self._flush(pending)
            self._flush(pending)
        except Exception as exc:     # surfaced by the next submit() / close()
            self._error = exc
            self._free.release()

    def _flush(self, pending):
        if pending:
            with open(self.path, 'a', encoding='utf-8') as f:
                f.write('\n'.join(pending) + '\n')
            self.lines_written += len(pending)
            pending.clear()

    def close(self):
        self._q.put(None)
        self._thread.join()
        if self._error is not None:
            raise self._error

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self.close()
        return False
