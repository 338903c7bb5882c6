"""vstar_amd — MI355X-native engine for the V* guided-visual-search hot path.

Drop-in surfaces (lazy imports: nothing heavy is loaded until used):
    vstar_amd.VSM, vstar_amd.visual_search ............. visual_search.py (wrapper class + crop scheduler)
    vstar_amd.VQA_LLM .................................. vstar_bench_eval.py (SEAL VQA-LLM wrapper class)
    vstar_amd.VSMForCausalLM, load_pretrained_model .... the reference's model-loading API (vstar_amd/api.py)
"""
_LAZY = {
    "VSM": ("vstar_amd.vsm", "VSM"), "visual_search": ("vstar_amd.search", "visual_search"),
    "VQA_LLM": ("vstar_amd.vqa", "VQA_LLM"), "VSMForCausalLM": ("vstar_amd.api", "VSMForCausalLM"),
    "load_pretrained_model": ("vstar_amd.api", "load_pretrained_model"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(mod), attr)
    raise AttributeError(f"module 'vstar_amd' has no attribute {name!r}")
