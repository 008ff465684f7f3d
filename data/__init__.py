"""Import-level drop-in for the reference's top-level ``data`` package: ``data.{cvs,proc,challenge}.config_*`` (``load_config``)
and the challenge / proc dataset builders.  The reference's data FILES are not shipped; point ``config.data_path`` at them."""
