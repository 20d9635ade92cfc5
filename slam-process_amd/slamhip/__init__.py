"""slamhip: MI355X (gfx950) implementation of the slam-process decode -> correct -> raster path.

`slamhip.ops` wraps the host-pointer C-ABI calls with NumPy arrays, `slamhip.pipeline` the batched
device-resident path, `slamhip.batch` whole captures in one ragged batch, `slamhip.dist` the sharding and the gather
of tiles, `slamhip.run_logs` the command line over both, `slamhip.synth` the synthetic capture generator,
`slamhip.xlsx` / `slamhip.png` the workbook and image files.  Importing the package does not touch the GPU.
"""
__all__ = ["ops", "pipeline", "batch", "dist", "run_logs", "synth", "xlsx", "png"]
