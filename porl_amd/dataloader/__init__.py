from .dataloader import DeviceDataset, EpochLoader, pack_csv_dir, pack_rows  # noqa: F401
