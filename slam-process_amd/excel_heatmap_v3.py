"""Drop-in for the reference's excel_heatmap_v3.py: corrected rows -> BS/UE mean-RSSI heatmap PNG.

Same module-level names, arguments, defaults, return types, exceptions and CLI flags as the reference
script (excel_heatmap_v3.py:36-62 logger, :68-184 main-data reader, :187-258 mapping reader, :264-280
map_angles, :286-314 compute_rssi_matrix, :320-368 blur, :374-475 generate_heatmap, :481-552 driver,
:558-646 CLI).  The mean pivot, the NaN-aware blur, the colour scale and the cell colours come from the
GPU through libslamhip.so; matplotlib only draws the figure furniture (axes, colour bar, title, grid)
around the device-coloured mesh.  There is no CPU implementation behind the compute functions.

What differs from the reference: workbooks are read with the standard library (no openpyxl);
generate_heatmap colours matrices of up to 256 x 256 cells (the library's axis limit; beams are 0..63,
so a mapped table has at most 64 angles an axis).  RSS may be any float: whole numbers are summed
exactly, fractions with pandas' compensated sum in row order.
"""
from __future__ import annotations

import argparse
import logging
import math
import os
import sys
from datetime import datetime
from typing import Dict, List, Optional, Tuple

import matplotlib
import matplotlib.pyplot as plt
import numpy as np
import pandas as pd
from matplotlib.cm import ScalarMappable
from matplotlib.colors import LogNorm, Normalize

from slamhip import ops, xlsx


def _setup_chinese_font():
    for f in ("Microsoft YaHei", "SimHei", "MS Gothic"):
        try:
            plt.rcParams["font.sans-serif"] = [f]
            plt.rcParams["axes.unicode_minus"] = False
            return
        except Exception:
            continue
    plt.rcParams["axes.unicode_minus"] = False


def setup_logger(log_path: str) -> logging.Logger:
    logger = logging.getLogger("excel_heatmap")
    logger.setLevel(logging.INFO)
    if logger.handlers:
        return logger
    fmt = logging.Formatter(fmt="%(asctime)s | %(levelname)s | %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
    fh = logging.FileHandler(log_path, encoding="utf-8")
    fh.setLevel(logging.INFO)
    fh.setFormatter(fmt)
    sh = logging.StreamHandler(sys.stdout)
    sh.setLevel(logging.INFO)
    sh.setFormatter(fmt)
    logger.addHandler(fh)
    logger.addHandler(sh)
    return logger


def validate_beam_id(beam: int) -> bool:
    return isinstance(beam, (int, np.integer)) and 0 <= int(beam) <= 63


def _read_sheets(path: str, sheet_name: Optional[str], what: str):
    """One DataFrame when a sheet is named, else {sheet: DataFrame} like pd.read_excel(sheet_name=None)."""
    def frame(table):          # an empty cell is NaN, as pd.read_excel gives it, also in a column that holds text
        return pd.DataFrame({k: [np.nan if v is None else v for v in col] for k, col in table.items()})

    try:
        if sheet_name is not None:
            return frame(xlsx.read_table(path, sheet_name))
        return {n: frame(t) for n, t in xlsx.read_all_tables(path).items()}
    except Exception as e:
        raise RuntimeError(f"读取{what}Excel失败: {e}")


def _choose_sheet(df_obj, needed: List[str], sheet_name: Optional[str], logger, label: str) -> pd.DataFrame:
    if not isinstance(df_obj, dict):
        if logger and sheet_name is not None:
            logger.info(f"使用{label}工作表: {sheet_name}")
        return df_obj
    chosen = None
    for name, d in df_obj.items():
        if all(c in d.columns for c in needed):
            chosen = name
            break
    if chosen is None:
        chosen = next(iter(df_obj.keys()))
    if logger:
        logger.info(f"使用{label}工作表: {chosen}")
    return df_obj[chosen]


def _int_or_nan(x):
    try:
        return int(x)
    except Exception:
        return np.nan


def _float_or_nan(x):
    try:
        return float(x)
    except Exception:
        return np.nan


def read_main_data(input_path: str, ue_col: str = "UE_Beam", bs_col: str = "BS_Beam", rss_col: str = "RSS值",
                   flag_col: Optional[str] = None, sheet_name: Optional[str] = None,
                   logger: Optional[logging.Logger] = None) -> pd.DataFrame:
    """Workbook -> DataFrame(ue_beam, bs_beam, rss[, flag]) of clean rows with beams in 0..63 (reference :75-184)."""
    if logger:
        logger.info(f"开始读取主数据文件: {input_path}")
    if not os.path.exists(input_path):
        raise FileNotFoundError(f"主数据文件不存在: {input_path}")
    needed = [ue_col, bs_col, rss_col] + ([flag_col] if flag_col else [])
    df = _choose_sheet(_read_sheets(input_path, sheet_name, "主数据"), needed, sheet_name, logger, "主数据")
    for col in needed:
        if col not in df.columns:
            raise KeyError(f"主数据缺少必要列: {col}")
    names = {ue_col: "ue_beam", bs_col: "bs_beam", rss_col: "rss"}
    if flag_col:
        names[flag_col] = "flag"
    df = df[needed].rename(columns=names)
    df["ue_beam"] = df["ue_beam"].apply(_int_or_nan)
    df["bs_beam"] = df["bs_beam"].apply(_int_or_nan)
    if "flag" in df.columns:
        df["flag"] = df["flag"].apply(_int_or_nan)
    df["rss"] = df["rss"].apply(_float_or_nan)
    before = len(df)
    df = df.dropna(subset=["ue_beam", "bs_beam", "rss"] + (["flag"] if "flag" in df.columns else []))
    after_dropna = len(df)
    # validate_beam_id accepts integers only: a beam column that held an empty or non-numeric cell has become a
    # float column, and then none of its rows passes (reference :166-168)
    ok = pd.Series(True, index=df.index)
    for c in ("ue_beam", "bs_beam"):
        ok &= df[c].between(0, 63) if df[c].dtype.kind in "iu" else False
    if "flag" in df.columns:
        ok &= df["flag"] == 1
    df = df[ok]
    if logger:
        logger.info(f"原始行数: {before}, 去NaN后: {after_dropna}, 过滤FLAG与波束有效后: {len(df)}")
    if df.empty:
        raise ValueError("清洗后主数据为空，请检查数据格式是否正确")
    df = df.copy()
    df["ue_beam"] = df["ue_beam"].astype(np.int64)
    df["bs_beam"] = df["bs_beam"].astype(np.int64)
    return df


def read_mapping(map_path: str, id_col: str = "BeamID", angle_col: str = "Angle", sheet_name: Optional[str] = None,
                 logger: Optional[logging.Logger] = None) -> Dict[int, float]:
    """Workbook -> {BeamID 0..63: angle in degrees} (reference :187-258)."""
    if logger:
        logger.info(f"开始读取映射文件: {map_path}")
    if not os.path.exists(map_path):
        raise FileNotFoundError(f"映射文件不存在: {map_path}")
    df = _choose_sheet(_read_sheets(map_path, sheet_name, "映射"), [id_col, angle_col], sheet_name, logger, "映射")
    for col in (id_col, angle_col):
        if col not in df.columns:
            raise KeyError(f"映射文件缺少必要列: {col}")
    mapping: Dict[int, float] = {}
    for bid, ang in zip(df[id_col].tolist(), df[angle_col].tolist()):
        try:
            b, a = int(bid), float(ang)
        except Exception:
            continue
        if 0 <= b <= 63:
            mapping[b] = a
    if logger:
        logger.info(f"映射条目数: {len(mapping)} (期望: 64)")
        missing = [i for i in range(64) if i not in mapping]
        if missing:
            logger.warning(f"映射缺失 BeamID: {missing}")
    if not mapping:
        raise ValueError("映射数据为空或格式错误")
    return mapping


def map_angles(df: pd.DataFrame, mapping: Dict[int, float], logger: Optional[logging.Logger] = None) -> pd.DataFrame:
    """Adds AoA (from ue_beam) and AoD (from bs_beam); rows with an unmapped beam go (reference :264-280)."""
    df = df.copy()
    df["AoA"] = df["ue_beam"].map(mapping)
    df["AoD"] = df["bs_beam"].map(mapping)
    before = len(df)
    df = df.dropna(subset=["AoA", "AoD"])
    if logger:
        logger.info(f"角度映射后有效行数: {len(df)} / {before}")
    if df.empty:
        raise ValueError("角度映射后数据为空，可能映射表与数据不一致")
    return df


def compute_rssi_matrix(df: pd.DataFrame, logger: Optional[logging.Logger] = None) -> Tuple[np.ndarray, List[float], List[float]]:
    """Mean RSS per (AoD, AoA) over the angles that occur, NaN where a pair never occurs (reference :286-314)."""
    aod_vals = np.unique(df["AoD"].to_numpy(np.float64))
    aoa_vals = np.unique(df["AoA"].to_numpy(np.float64))
    rss = df["rss"].to_numpy(np.float64)
    r_idx = np.searchsorted(aod_vals, df["AoD"].to_numpy(np.float64))
    c_idx = np.searchsorted(aoa_vals, df["AoA"].to_numpy(np.float64))
    ident = np.arange(64, dtype=np.int16)
    # whole numbers (every capture's RSS): exact integer sums, which is also what pandas' sum comes to; fractions (dBm
    # readings): pandas' compensated sum in row order, followed literally on the device
    whole = bool(np.all(rss == np.round(rss)) and np.all(np.abs(rss) < 2.0 ** 31))
    matrix = np.full((len(aod_vals), len(aoa_vals)), np.nan)
    # The device pivot works on 64 x 64 index pairs (beams are 0..63, so a mapped table has at most 64 angles an axis and
    # this loop runs once); a table that was given more angles directly is served 64 x 64 block by block, each row going
    # to the block its pair of indices lies in, in row order.
    for br in range(0, len(aod_vals), 64):
        for bc in range(0, len(aoa_vals), 64):
            if len(aod_vals) <= 64 and len(aoa_vals) <= 64:
                rr, cc, vv = r_idx, c_idx, rss
            else:
                sel = (r_idx // 64 == br // 64) & (c_idx // 64 == bc // 64)
                if not sel.any():
                    continue
                rr, cc, vv = r_idx[sel] - br, c_idx[sel] - bc, rss[sel]
            if whole:
                _s, _c, mean = ops.pivot_mean(cc, rr, vv.astype(np.int64), ident, 64)
            else:
                _s, _c, mean = ops.pivot_mean_f64(cc, rr, vv, ident, 64)
            h, w = min(64, len(aod_vals) - br), min(64, len(aoa_vals) - bc)
            matrix[br:br + h, bc:bc + w] = mean[:h, :w]
    if logger:
        logger.info(f"RSSI矩阵尺寸: {matrix.shape} (AoD: {len(aod_vals)} x AoA: {len(aoa_vals)})")
    return matrix, aod_vals.tolist(), aoa_vals.tolist()


def gaussian_kernel(sigma: float) -> np.ndarray:
    """Odd-sized normalised Gaussian of about 6 sigma (reference :320-333); [[1.0]] when sigma <= 0.  A few dozen
    numbers worked out on the host, as the library works out its own copy before a blur; the blur itself
    (gaussian_blur_nan_aware) runs on the device."""
    if sigma <= 0:
        return np.ones((1, 1), dtype=np.float64)
    size = max(3, int(math.ceil(6 * sigma)))
    size += 1 - size % 2
    r = np.arange(size) - size // 2
    k = np.exp(-(r[None, :] ** 2 + r[:, None] ** 2) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(np.float64)


def gaussian_blur_nan_aware(data: np.ndarray, sigma: float) -> np.ndarray:
    """Normalised convolution that ignores NaN cells, edge padding (reference :336-368)."""
    if sigma <= 0:
        return data
    return ops.blur_nan(np.asarray(data, dtype=np.float64), float(sigma))


def _edges(vals: List[float]) -> np.ndarray:
    v = np.asarray(vals, dtype=np.float64)
    if len(v) == 1:
        return np.array([v[0] - 0.5, v[0] + 0.5])
    mid = (v[:-1] + v[1:]) / 2.0
    return np.concatenate([[v[0] - (v[1] - v[0]) / 2.0], mid, [v[-1] + (v[-1] - v[-2]) / 2.0]])


def _lut_of(colormap: str) -> np.ndarray:
    """256 x 3 bytes of a matplotlib colormap as the Agg renderer writes them (rounded): entry i is the colour of the
    normalised value i / 256, which for the usual 256-entry maps is the map's own entry i and for a map of fewer
    colours (tab10, Set1, ...) the colour that value falls on.  The device looks up entry floor(norm * 256), so for a
    map whose number of colours N does not divide 256 a cell whose normalised value lies in [k / N, ceil(256 k / N) / 256)
    -- a band of less than 1 / 256 of the range above each colour boundary -- takes the colour below the boundary where
    matplotlib takes the one above: exact for maps of 256, 128, 64 ... colours (viridis, the default, among them), not
    pinned inside those bands for the others."""
    cmap = plt.get_cmap(colormap)
    return np.rint(np.asarray(cmap(np.arange(256) / 256.0))[:, :3] * 255).astype(np.uint8)


def generate_heatmap(matrix: np.ndarray, aod_list: List[float], aoa_list: List[float], output_path: str, title: str,
                     colormap: str = "viridis", use_log: bool = True, vmin: Optional[float] = None,
                     vmax: Optional[float] = None, blur_sigma: float = 0.0,
                     logger: Optional[logging.Logger] = None) -> None:
    """Blur, scale, colour and save the 10 x 8 in figure at dpi 150 (reference :374-475)."""
    _setup_chinese_font()
    mat = np.array(matrix, dtype=np.float64, copy=True)
    if blur_sigma and blur_sigma > 0:
        if logger:
            logger.info(f"对矩阵进行高斯模糊，sigma={blur_sigma}")
        mat = gaussian_blur_nan_aware(mat, blur_sigma)
    try:
        cells, _tile, lo, hi = ops.render(mat, aod_list, aoa_list, _lut_of(colormap), use_log=use_log, vmin=vmin,
                                          vmax=vmax, want_tile=False)
    except ops._lib.SlamError as e:
        if e.code == ops._lib.E_ALLNAN:
            raise ValueError("矩阵数据全为NaN，无法生成热力图")
        if e.code == ops._lib.E_LIMITS:
            raise ValueError("Invalid vmin or vmax")           # what the reference's LogNorm says (:415-418)
        raise
    fig, ax = plt.subplots(figsize=(10, 8), dpi=120)
    ax.pcolormesh(_edges(aoa_list), _edges(aod_list), cells.astype(np.float64) / 255.0, shading="flat")
    norm = LogNorm(vmin=lo, vmax=hi) if use_log else Normalize(vmin=lo, vmax=hi)
    cbar = fig.colorbar(ScalarMappable(norm=norm, cmap=plt.get_cmap(colormap)), ax=ax)
    cbar.set_label("RSSI强度" + ("（对数刻度）" if use_log else "（线性刻度）"))
    ax.set_title(title)
    ax.set_xlabel("AoA（UE侧，度）")
    ax.set_ylabel("AoD（BS侧，度）")
    ax.grid(True, linestyle="--", alpha=0.2)
    out_dir = os.path.dirname(output_path)
    if out_dir and not os.path.exists(out_dir):
        os.makedirs(out_dir, exist_ok=True)
    plt.tight_layout()
    plt.savefig(output_path, dpi=150)
    plt.close(fig)
    if logger:
        logger.info(f"热力图已保存: {output_path}")


def process_excel_to_heatmap(input_path: str, map_path: str, output_path: Optional[str] = None,
                             sheet_name_main: Optional[str] = None, sheet_name_map: Optional[str] = None,
                             colormap: str = "viridis", use_log: bool = True, vmin: Optional[float] = None,
                             vmax: Optional[float] = None, blur_sigma: float = 0.0,
                             logger: Optional[logging.Logger] = None) -> str:
    """read -> validate -> map -> pivot -> draw; returns the PNG path (reference :481-552)."""
    where = os.path.dirname(output_path or input_path)
    lg = logger or setup_logger(os.path.join(where, "excel_heatmap_%s.log" % datetime.now().strftime("%Y%m%d_%H%M%S")))
    try:
        for line in ("====== 开始处理Excel数据并生成热力图 ======", f"主数据文件: {input_path}", f"映射文件: {map_path}"):
            lg.info(line)
        rows = read_main_data(input_path=input_path, sheet_name=sheet_name_main, logger=lg)
        rows = map_angles(rows, read_mapping(map_path=map_path, sheet_name=sheet_name_map, logger=lg), logger=lg)
        grid = compute_rssi_matrix(rows, logger=lg)
        name = os.path.basename(input_path)
        if not (output_path or "").strip():
            output_path = os.path.join(os.path.dirname(input_path), "heatmap_outputs", os.path.splitext(name)[0] + "_heatmap.png")
        generate_heatmap(*grid, output_path, f"BS-UE 波束对平均RSSI热力图 ({name})", colormap, use_log, vmin, vmax,
                         blur_sigma, lg)
        lg.info("====== 处理完成 ======")
        return output_path
    except Exception as e:
        lg.error(f"处理失败: {e}", exc_info=True)
        raise


# The reference's options with its defaults (:558-604); the help texts are this build's own wording.
_OPTIONS = (
    ("--input", str, r"D:\Code\SLAMPro\UE21\debugDoc\Serial Debug 2026-01-26 165358_filtered.xlsx", "修正后的主数据工作簿 (xlsx)"),
    ("--mapping", str, r"D:\桌面\SLAMM\beam_angle.xlsx", "波束号到角度的对照工作簿 (xlsx)"),
    ("--output", str, "", "PNG 输出位置；留空则写到主数据旁的 heatmap_outputs 目录"),
    ("--sheet-main", str, None, "主数据所在工作表"),
    ("--sheet-map", str, None, "对照表所在工作表"),
    ("--colormap", str, "viridis", "matplotlib 色图名称"),
    ("--vmin", float, None, "对数色标下限（线性色标不使用）"),
    ("--vmax", float, None, "对数色标上限（线性色标不使用）"),
    ("--blur-sigma", float, 1.0, "高斯平滑的 sigma，0 表示不平滑"),
)


def build_arg_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="修正后的工作簿 -> BS/UE 波束对平均 RSSI 热力图 (GPU)")
    for flag, kind, default, text in _OPTIONS:
        p.add_argument(flag, type=kind, default=default, help=text)
    p.add_argument("--logscale", action="store_true", default=False, help="改用对数色标")
    return p


def main():
    args = build_arg_parser().parse_args()
    if args.output:
        log_dir = os.path.dirname(args.output)
    else:
        log_dir = os.path.dirname(args.input) if os.path.isfile(args.input) else os.getcwd()
    logger = setup_logger(os.path.join(log_dir, "excel_heatmap_%s.log" % datetime.now().strftime("%Y%m%d_%H%M%S")))
    shown = [(k, getattr(args, k)) for k in ("input", "mapping", "output", "sheet_main", "sheet_map", "colormap",
                                             "logscale", "vmin", "vmax", "blur_sigma")]
    logger.info("参数: " + ", ".join(f"{k}={v}" if k in ("logscale", "vmin", "vmax", "blur_sigma") else f"{k}='{v}'"
                                    for k, v in shown))
    try:
        png_path = process_excel_to_heatmap(args.input, args.mapping, args.output or None, args.sheet_main, args.sheet_map,
                                            args.colormap, args.logscale, args.vmin, args.vmax, args.blur_sigma, logger)
        logger.info(f"输出PNG: {png_path}")
    except Exception as e:
        logger.error(f"执行失败: {e}", exc_info=True)
        sys.exit(1)


if __name__ == "__main__":
    main()
