"""Offline tooling for the regrid path: the npz files the ``regrid`` filter consumes.

The reference ships ``anemoi-transform make-regrid-file {mir-matrix, global-on-lam-mask}`` and
``get-grid`` (R: commands/make-regrid-file.py:245-275, commands/get-grid.py:16-52); ``mir-matrix`` shells out
to ECMWF's ``mir`` binary, which is not available offline.  This CLI writes the SAME file formats
(R: make-regrid-file.py:150-160 matrix npz, :240 mask npz, get-grid.py:52 grid npz) from the builders of this
package, so recipes such as ``regrid: {matrix: file.npz}`` / ``regrid: {mask: file.npz}`` run unchanged:

    atx make-regrid-file knn-matrix      SOURCE TARGET --k 4 --output m.npz [--device]
    atx make-regrid-file bilinear-matrix o96 1.0 --output m.npz          (octahedral sources)
    atx make-regrid-file global-on-lam-mask GLOBAL LAM --distance-km 10 --output mask.npz [--device]
    atx get-grid o1280 --output grid-o1280.npz
    atx filters list

Grids are names understood by ``grids.lookup`` (O<N>, F<N>, lat-lon increments) or ``.npz`` files with
``latitudes`` / ``longitudes``.
"""

from __future__ import annotations

import argparse
import sys

import numpy as np


def _grid(spec: str):
    from .grids import lookup

    return lookup(spec)


def _rows(spec: str):
    """Row structure of a formula source grid (O<N>, F<N>, N320-sized, regular lat-lon increments) for the bilinear builder."""
    from .grids import row_structure

    rows = row_structure(spec)
    if rows is None:
        raise SystemExit(f"bilinear-matrix needs a row-structured formula source grid (O<N>, F<N>, n320-sized, dlat/dlon), got {spec!r}")
    return rows


def main(argv: list[str] | None = None) -> int:
    parser = argparse.ArgumentParser(prog="atx", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = parser.add_subparsers(dest="command", required=True)

    mk = sub.add_parser("make-regrid-file", help="write a regrid matrix / mask npz").add_subparsers(dest="kind", required=True)
    knn = mk.add_parser("knn-matrix", help="k-NN inverse-distance weights (k = 1: nearest neighbour)")
    knn.add_argument("source_grid")
    knn.add_argument("target_grid")
    knn.add_argument("--k", type=int, default=4)
    knn.add_argument("--device", action="store_true", help="neighbour search on the GPU (atx_knn_*)")
    knn.add_argument("--output", required=True)
    bil = mk.add_parser("bilinear-matrix", help="bilinear weights from an octahedral grid")
    bil.add_argument("source_grid")
    bil.add_argument("target_grid")
    bil.add_argument("--output", required=True)
    gol = mk.add_parser("global-on-lam-mask", help="indices of the global points near a limited-area grid")
    gol.add_argument("global_grid")
    gol.add_argument("lam_grid")
    gol.add_argument("--distance-km", type=float, default=None)
    gol.add_argument("--device", action="store_true")
    gol.add_argument("--output", required=True)

    gg = sub.add_parser("get-grid", help="write a formula grid as npz")
    gg.add_argument("grid")
    gg.add_argument("--output", required=True)

    fl = sub.add_parser("filters", help="registered filters").add_subparsers(dest="action", required=True)
    fl.add_parser("list")

    args = parser.parse_args(argv)

    if args.command == "get-grid":
        g = _grid(args.grid)
        np.savez(args.output, latitudes=g["latitudes"], longitudes=g["longitudes"])  # R: commands/get-grid.py:52
        return 0

    if args.command == "filters":
        from .filters import filter_registry

        for name in filter_registry.registered:  # R: commands/filters.py:45
            aliases = filter_registry.aliases().get(name)
            print(name + (f" (aliases: {', '.join(aliases)})" if aliases else ""))
        return 0

    from . import interp, spatial

    if args.kind == "knn-matrix":
        src, tgt = _grid(args.source_grid), _grid(args.target_grid)
        idx, w = interp.knn_inverse_distance(src, tgt, k=args.k, device=args.device)
        interp.save_matrix_npz(args.output, interp.ell_to_csr(idx, w, len(src["latitudes"])), src, tgt)
    elif args.kind == "bilinear-matrix":
        src, tgt = _grid(args.source_grid), _grid(args.target_grid)
        interp.save_matrix_npz(args.output, interp.bilinear_rows(*_rows(args.source_grid), tgt), src, tgt)
    else:
        glob, lam = _grid(args.global_grid), _grid(args.lam_grid)
        mask = spatial.global_on_lam_mask(np.asarray(lam["latitudes"]), np.asarray(lam["longitudes"]), np.asarray(glob["latitudes"]),
                                          np.asarray(glob["longitudes"]), distance_km=args.distance_km, device=args.device)
        np.savez(args.output, mask=mask)  # R: commands/make-regrid-file.py:240
    return 0


if __name__ == "__main__":
    sys.exit(main())
