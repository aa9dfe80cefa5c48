"""Drop-in for the reference's ``utils/MinkowskiCollate.py`` (minkowski_collate_fn, :4-15) without MinkowskiEngine."""
import torch


def sparse_collate(coords, feats, labels=None, dtype=torch.int32, device=None):
    """What ``ME.utils.sparse_collate`` does for this path: batch index in column 0, features and labels
    concatenated; float coordinates are floored only when ``dtype`` is int32."""
    N = sum(int(c.shape[0]) for c in coords)
    if N != sum(int(f.shape[0]) for f in feats):
        raise ValueError("coordinates and features disagree on the number of points")
    D = coords[0].shape[1]
    bcoords = torch.zeros((N, D + 1), dtype=dtype, device=device if device is not None else coords[0].device)
    s = 0
    for b, c in enumerate(coords):
        if dtype == torch.int32 and c.dtype in (torch.float32, torch.float64):
            c = c.floor()
        n = int(c.shape[0])
        bcoords[s:s + n, 1:] = c
        bcoords[s:s + n, 0] = b
        s += n
    feats_batch = torch.cat(list(feats), 0)
    if labels is None:
        return bcoords, feats_batch
    return bcoords, feats_batch, torch.cat(list(labels), 0)


def minkowski_collate_fn(list_data):
    coordinates_batch, features_batch, labels_batch = sparse_collate(
        [d["coordinates"] for d in list_data],
        [d["features"] for d in list_data],
        [d["label"].reshape(1) for d in list_data],
        dtype=torch.float32,
    )
    return {"coordinates": coordinates_batch, "features": features_batch, "labels": labels_batch,
            "num_clouds": len(list_data)}
