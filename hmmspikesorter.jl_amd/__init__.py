"""hmmspikesorter.jl_amd -- MI355X-native HMM spike-sorting hot path (forward/backward/update,
viterbi, reconstruct_signal of grero/HMMSpikeSorter.jl) behind libhmmsort_hip.so.

The directory name is the one the build contract fixes; it is not a valid Python identifier, so
import it through the root-level shim:  `import hmmsort_amd`.
"""
from . import _lib, dist, synth
from ._lib import (ENGINE_AUTO, ENGINE_BLOCKED, ENGINE_RING, ENGINE_STRICT, ENGINE_WAVE, HmmsortError, device_count,
                   get_option, set_option, shutdown)
from .api import (HMMSpikeTemplateModel, HMMSpikingModel, StateMatrix, backward, extract_spiketimes,
                  fit, forward,
                  predict, reconstruct_signal, train_model, train_step, unroll_mlseq, update,
                  viterbi)
from .device import Plan
from .postprocess import (condense_candidates, condense_templates, find_best_overlap, match_templates,
                          prune_templates, remove_small, remove_sparse)
from .sortdata import get_lp, sort_data
from .synth import create_signal, create_spike_template

__all__ = ["StateMatrix", "HMMSpikeTemplateModel", "HMMSpikingModel", "forward", "backward",
           "update", "train_model", "train_step", "viterbi", "reconstruct_signal", "unroll_mlseq",
           "fit", "predict", "extract_spiketimes", "Plan", "create_signal", "create_spike_template", "HmmsortError",
           "set_option", "get_option", "shutdown", "device_count", "ENGINE_AUTO", "ENGINE_STRICT",
           "ENGINE_RING", "ENGINE_BLOCKED", "ENGINE_WAVE", "get_lp", "sort_data", "find_best_overlap",
           "condense_candidates", "condense_templates", "remove_sparse", "remove_small", "prune_templates",
           "match_templates"]
