from .ops import interpolate, interpolate_var_size_batch, lengths, lengths_var_size_batch

__all__ = ["interpolate", "interpolate_var_size_batch", "lengths", "lengths_var_size_batch"]
