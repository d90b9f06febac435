from .functional import (QUANT_TYPE, get_quant_type_from_children_dict, quantize_nf4, quantize_state_dict,
                         replace_by_prequantized_weights, replace_to_quant_linear, validate_quant_type)
from .nf4 import NF4Linear

__all__ = ["QUANT_TYPE", "NF4Linear", "quantize_nf4", "quantize_state_dict", "replace_by_prequantized_weights",
           "replace_to_quant_linear", "validate_quant_type", "get_quant_type_from_children_dict"]
