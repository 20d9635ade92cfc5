"""slamhip: MI355X (gfx950) implementation of the slam-process decode -> correct -> raster path.

`slamhip.ops` wraps the host-pointer C-ABI calls with NumPy arrays, `slamhip.pipeline` the batched
device-resident path, `slamhip.synth` the synthetic capture generator, `slamhip.xlsx` the workbook
reader/writer.  Importing the package does not touch the GPU.
"""
__all__ = ["ops", "synth", "xlsx"]
