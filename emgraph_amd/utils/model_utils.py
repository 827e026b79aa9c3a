"""save_model / restore_model with the reference's pickle schema (utils/model_utils.py:63-87,142-154):
{class_name, hyperparams, is_fitted, ent_to_idx, rel_to_idx, is_calibrated, model_params=[ent_emb, rel_emb],
large_graph, calibration_parameters}; model_params are numpy arrays."""
import glob
import importlib
import logging
import pickle
from time import gmtime, strftime

logger = logging.getLogger(__name__)

DEFAULT_MODEL_NAMES = "{0}.model.pkl"


def save_model(model, model_name_path=None, protocol=pickle.HIGHEST_PROTOCOL):
    obj = {
        "class_name": model.__class__.__name__,
        "hyperparams": model.all_params,
        "is_fitted": model.is_fitted,
        "ent_to_idx": model.ent_to_idx,
        "rel_to_idx": model.rel_to_idx,
        "is_calibrated": model.is_calibrated,
    }
    model.get_embedding_model_params(obj)
    if model_name_path is None:
        model_name_path = DEFAULT_MODEL_NAMES.format(strftime("%Y_%m_%d-%H_%M_%S", gmtime()))
    with open(model_name_path, "wb") as fw:
        pickle.dump(obj, fw, protocol=protocol)


def restore_model(model_name_path=None):
    if model_name_path is None:
        logger.warning("There is no model name specified. We will try to lookup the latest default saved model...")
        default_models = glob.glob("*.model.pkl")
        if len(default_models) == 0:
            raise Exception("No default model found. Please specify model_name_path...")
        model_name_path = default_models[len(default_models) - 1]
    try:
        with open(model_name_path, "rb") as fr:
            restored_obj = pickle.load(fr)
        module = importlib.import_module("emgraph_amd.models")
        class_ = getattr(module, restored_obj["class_name"])
        model = class_(**restored_obj["hyperparams"])
        model.is_fitted = restored_obj["is_fitted"]
        model.ent_to_idx = restored_obj["ent_to_idx"]
        model.rel_to_idx = restored_obj["rel_to_idx"]
        model.is_calibrated = restored_obj.get("is_calibrated", False)
        model.restore_model_params(restored_obj)
    except pickle.UnpicklingError as e:
        msg = "Error unpickling model {} : {}.".format(model_name_path, e)
        logger.debug(msg)
        raise Exception(msg)
    except (IOError, FileNotFoundError):
        msg = "No model found: {}.".format(model_name_path)
        logger.debug(msg)
        raise FileNotFoundError(msg)
    return model
