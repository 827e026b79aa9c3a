"""save_model / restore_model writing and reading the reference's pickle layout (emgraph/utils/model_utils.py:63-87,
142-154), so that model files are interchangeable: one dict with class_name, hyperparams, is_fitted, ent_to_idx,
rel_to_idx, is_calibrated plus what the model adds itself (model_params = [entity table, relation table] as numpy
arrays, large_graph, calibration_parameters)."""
import glob
import importlib
import logging
import pickle
import time

logger = logging.getLogger(__name__)

DEFAULT_MODEL_NAMES = "{0}.model.pkl"
_HEADER = ("class_name", "hyperparams", "is_fitted", "ent_to_idx", "rel_to_idx", "is_calibrated")


def save_model(model, model_name_path=None, protocol=pickle.HIGHEST_PROTOCOL):
    """Pickle ``model``.  Without a path the file is named after the current UTC time, <stamp>.model.pkl."""
    payload = dict(zip(_HEADER, (type(model).__name__, model.all_params, model.is_fitted, model.ent_to_idx,
                                 model.rel_to_idx, model.is_calibrated)))
    model.get_embedding_model_params(payload)          # the model appends its tables and flags
    target = model_name_path or DEFAULT_MODEL_NAMES.format(time.strftime("%Y_%m_%d-%H_%M_%S", time.gmtime()))
    with open(target, "wb") as sink:
        pickle.dump(payload, sink, protocol=protocol)


def _latest_default_model():
    logger.warning("There is no model name specified. We will try to lookup the latest default saved model...")
    found = glob.glob(DEFAULT_MODEL_NAMES.format("*"))
    if not found:
        raise Exception("No default model found. Please specify model_name_path...")
    return found[-1]


def restore_model(model_name_path=None):
    """Rebuild the model object a ``save_model`` file describes (FileNotFoundError if there is no such file)."""
    path = model_name_path if model_name_path is not None else _latest_default_model()
    try:
        with open(path, "rb") as source:
            stored = pickle.load(source)
    except pickle.UnpicklingError as err:
        text = "Error unpickling model {} : {}.".format(path, err)
        logger.debug(text)
        raise Exception(text)
    except (IOError, FileNotFoundError):
        text = "No model found: {}.".format(path)
        logger.debug(text)
        raise FileNotFoundError(text)
    model_class = getattr(importlib.import_module("emgraph_amd.models"), stored["class_name"])
    model = model_class(**stored["hyperparams"])
    for field in ("is_fitted", "ent_to_idx", "rel_to_idx"):
        setattr(model, field, stored[field])
    model.is_calibrated = stored.get("is_calibrated", False)
    model.restore_model_params(stored)
    return model
