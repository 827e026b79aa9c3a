"""Data adapters on the caller side of the hot path — host-side mirror of emgraph/datasets/
abstract_dataset_adapter.py and numpy_adapter.py (same class and method names, argument meaning and errors), so
that `fit(adapter)` / `evaluate_performance(adapter, ...)` calls written against the reference keep working.

What differs underneath: the reference's NumpyDatasetAdapter answers filter queries through a temporary SQLite
database (numpy_adapter.py:229-247 -> sqlite_adapter.py: one connect + two SQL queries per test triple); here
`set_filter` builds the vectorised `FilterIndex` (evaluation/ranking.py) once and the ranking path consumes it as
a CSR.  `get_next_batch(..., use_filter=True)` still yields the per-triple lists the reference yields, for callers
that iterate the adapter themselves.  Out of scope: SQLiteAdapter as a storage engine, OneToNDatasetAdapter.
"""
from __future__ import annotations

import abc

import numpy as np


class EmgraphBaseDatasetAdaptor(abc.ABC):
    """abstract_dataset_adapter.py:4-143 — the interface a data source offers to fit() / evaluate_performance()."""

    def __init__(self):
        self.dataset = {}
        self.rel_to_idx = {}
        self.ent_to_idx = {}
        self.mapped_status = {}
        self.focusE_numeric_edge_values = {}

    def use_mappings(self, rel_to_idx, ent_to_idx):
        """:44-60: adopt existing dictionaries; every stored dataset has to be mapped again."""
        self.rel_to_idx = rel_to_idx
        self.ent_to_idx = ent_to_idx
        for key in self.dataset.keys():
            self.mapped_status[key] = False

    def generate_mappings(self, use_all=False):
        raise NotImplementedError("Abstract Method not implemented!")

    def get_size(self, dataset_type="train"):
        raise NotImplementedError("Abstract Method not implemented!")

    def data_exists(self, dataset_type="train"):
        raise NotImplementedError("Abstract Method not implemented!")

    def set_data(self, dataset, dataset_type=None, mapped_status=False):
        raise NotImplementedError("Abstract Method not implemented!")

    def map_data(self, remap=False):
        raise NotImplementedError("Abstract Method not implemented!")

    def set_filter(self, filter_triples):
        raise NotImplementedError("Abstract Method not implemented!")

    def get_next_batch(self, batches_count=-1, dataset_type="train", use_filter=False):
        raise NotImplementedError("Abstract Method not implemented!")

    def cleanup(self):
        raise NotImplementedError("Abstract Method not implemented!")


class NumpyDatasetAdapter(EmgraphBaseDatasetAdaptor):
    """numpy_adapter.py:6-260."""

    def __init__(self):
        super().__init__()
        self.filter_adapter = None  # reference name of the filter backend; here the mapped filter triples
        self.filter_index = None    # evaluation.ranking.FilterIndex over them (what the ranking path consumes)

    def generate_mappings(self, use_all=False):
        """:19-44."""
        from ..evaluation.protocol import create_mappings
        if use_all:
            data = np.concatenate([self.dataset[key] for key in self.dataset.keys()], axis=0)
        else:
            data = self.dataset["train"]
        self.rel_to_idx, self.ent_to_idx = create_mappings(data)
        return self.rel_to_idx, self.ent_to_idx

    def get_size(self, dataset_type="train"):
        return self.dataset[dataset_type].shape[0]

    def data_exists(self, dataset_type="train"):
        return dataset_type in self.dataset.keys()

    def get_next_batch(self, batches_count=-1, dataset_type="train", use_filter=False):
        """Generator over the batches of one split (the protocol of numpy_adapter.py:79-131): rows in stored order, cut
        into ``batches_count`` contiguous pieces of ceil(n / batches_count) rows, as int32; ``batches_count = -1`` =
        one triple per batch.  Each item is a list: [triples] + [edge values of those rows, if the split has any]
        + [known objects of (s, p, ?), known subjects of (?, p, o)] when ``use_filter`` — the two lists shaped [n, 1], as
        the reference's SQL backend returns them."""
        if not self.mapped_status[dataset_type]:
            self.map_data()
        rows = self.dataset[dataset_type]
        n = self.get_size(dataset_type)
        step, pieces = (1, n) if batches_count == -1 else (int(np.ceil(n / batches_count)), batches_count)
        weights = self.focusE_numeric_edge_values.get(dataset_type)
        for lo in range(0, step * pieces, step):
            triples = np.int32(rows[lo:lo + step, :])
            item = [triples]
            if weights is not None:
                item.append(weights[lo:lo + step, :])
            if use_filter:
                item.extend(self.get_participating_entities(triples))
            yield item

    def get_participating_entities(self, x_triple):
        """what SQLiteAdapter.get_participating_entities (sqlite_adapter.py:449-508) returns for ONE triple:
        ({o} U known objects of (s, p, ?),  {s} U known subjects of (?, p, o)), each int [n, 1]."""
        from .. import _lib as L
        if self.filter_index is None:
            raise Exception("No filter has been set: call set_filter() first")
        x = np.asarray(x_triple).reshape(1, 3)
        n_ent = int(max(self.filter_index.max_entity, x[0, 0], x[0, 2])) + 1
        ptr, idx = self.filter_index.csr(x, L.EVAL_S_O, n_ent)
        return idx[ptr[0]:ptr[1]].reshape(-1, 1).astype(np.int64), idx[ptr[1]:ptr[2]].reshape(-1, 1).astype(np.int64)

    def map_data(self, remap=False):
        """Replace the labels of every split that still holds labels (all splits with ``remap``) by ids, creating the
        mappings from the train split first if there are none (numpy_adapter.py:133-154)."""
        from ..evaluation.protocol import to_idx
        if not self.rel_to_idx or not self.ent_to_idx:
            self.generate_mappings()
        pending = [name for name in self.dataset if remap is True or not self.mapped_status[name]]
        for name in pending:
            self.dataset[name] = to_idx(self.dataset[name], ent_to_idx=self.ent_to_idx, rel_to_idx=self.rel_to_idx)
            self.mapped_status[name] = True

    # what a split must look like: (predicate on the candidate, the reference's message for its failure)
    _SPLIT_CHECKS = (
        (lambda d: type(d) == np.ndarray, lambda d: "Invalid type for input data. Expected ndarray, got {}".format(type(d))),
        (lambda d: np.shape(d)[1] == 3,
         lambda d: "Invalid size for input data. Expected number of column 3, got {}".format(np.shape(d)[1])),
    )

    def _validate_data(self, data):
        """ValueError with the reference's message (numpy_adapter.py:156-179) unless ``data`` is an ndarray of 3 columns"""
        for holds, message in self._SPLIT_CHECKS:
            if not holds(data):
                raise ValueError(message(data))

    def set_data(self, dataset, dataset_type=None, mapped_status=False, focusE_numeric_edge_values=None):
        """Store one split (``dataset`` an array, ``dataset_type`` its name) or several (``dataset`` a dict name -> array,
        edge values then a dict too); ``mapped_status``: the arrays already hold ids.  With mappings present the new
        splits are mapped at once (numpy_adapter.py:181-227)."""
        if isinstance(dataset, dict):
            splits = dataset
            weights = focusE_numeric_edge_values or {}
        elif dataset_type is not None:
            splits = {dataset_type: dataset}
            weights = {} if focusE_numeric_edge_values is None else {dataset_type: focusE_numeric_edge_values}
        else:
            raise Exception("Incorrect usage. Expected a dictionary or a combination of dataset and it's type.")
        for name, rows in splits.items():
            self._validate_data(rows)
            self.dataset[name] = rows
            self.mapped_status[name] = mapped_status
            if focusE_numeric_edge_values is not None:
                self.focusE_numeric_edge_values[name] = weights[name]
        if self.rel_to_idx and self.ent_to_idx:
            self.map_data()

    def set_filter(self, filter_triples, mapped_status=False):
        """:229-247: the triples every test triple's corruptions are filtered against."""
        from ..evaluation.protocol import to_idx
        from ..evaluation.ranking import FilterIndex
        F = np.asarray(filter_triples)
        if not mapped_status:
            F = to_idx(F, ent_to_idx=self.ent_to_idx, rel_to_idx=self.rel_to_idx)
        self.filter_adapter = np.asarray(F, dtype=np.int64).reshape(-1, 3)
        self.filter_index = FilterIndex(self.filter_adapter)

    def cleanup(self):
        """:249-255."""
        self.filter_adapter = None
        self.filter_index = None


__all__ = ["EmgraphBaseDatasetAdaptor", "NumpyDatasetAdapter"]
