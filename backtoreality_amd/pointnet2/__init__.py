"""Drop-in `pointnet2` package: `pointnet2._ext` resolves here when the parent directory
(`backtoreality_amd/`) is on sys.path; `pointnet2_utils` / `pointnet2_modules` /
`pytorch_utils` import top-level when this directory is on sys.path (what the reference's
scripts do, e.g. train_Votenet_FSB.py:35-38).  See INTEGRATION.md."""
