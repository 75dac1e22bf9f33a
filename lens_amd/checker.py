"""Pre-flight checks (behaviour of lens/tools/checker.py:5-26): same assertions, same messages."""
import os


def check_args(args):
    base = os.path.join(args.data_dir, args.dataset, args.camera)
    assert os.path.exists(base), "Data directory does not exist: {}".format(base)
    ref_dir = os.path.join(base, args.reference)
    assert os.path.exists(ref_dir), "Reference directory does not exist: {}".format(ref_dir)
    inference = not args.train_model and not args.collect_data and not args.event_driven
    if inference:
        qry_dir = os.path.join(base, args.query)
        assert os.path.exists(qry_dir), "Query directory does not exist: {}".format(qry_dir)
    n_ref = len(os.listdir(ref_dir))
    assert args.reference_places * args.filter <= n_ref, \
        f"Not enough reference images for {args.reference_places} places and a filter of {args.filter}"
    if inference:
        n_qry = len(os.listdir(os.path.join(base, args.query)))
        assert args.query_places * args.filter <= n_qry, \
            f"Not enough query images for {args.query_places} places and a filter of {args.filter}"
    if args.collect_data:
        new_dir = os.path.join(base, args.data_name)
        assert not os.path.exists(new_dir), "Data directory already exists: {}".format(new_dir)
    # the reference builds a Conv2d(k, stride k) on a roi_dim x roi_dim tensor and checks its output size
    k = args.roi_dim // args.dims
    out_dim = (args.roi_dim - k) // k + 1 if k > 0 else 0
    assert out_dim == args.dims, "ROI dimension and final image size are incompatible"
