"""DataFeed: the {name: {"data", "attributes"}} dictionary handed to
DataManager.push_data_to_device (reference warp_drive/utils/data_feed.py:8-105)."""


class DataFeed(dict):
    def add_data(self, name, data, save_copy_and_apply_at_reset=False,
                 log_data_across_episode=False, **extra_attributes):
        attributes = {
            "save_copy_and_apply_at_reset": bool(save_copy_and_apply_at_reset),
            "log_data_across_episode": bool(log_data_across_episode),
        }
        attributes.update(extra_attributes)
        self[name] = {"data": data, "attributes": attributes}

    def add_data_list(self, data_list):
        """Entries are (name, data[, at_reset[, log]]) tuples or dicts with the add_data keys."""
        assert isinstance(data_list, list)
        for entry in data_list:
            assert len(entry) >= 2, "name and data are strictly required"
            if isinstance(entry, tuple):
                name, data = entry[0], entry[1]
                assert isinstance(name, str)
                flags = [f if isinstance(f, bool) else False for f in entry[2:4]]
                flags += [False] * (2 - len(flags))
                self.add_data(name, data, flags[0], flags[1])
            elif isinstance(entry, dict):
                self.add_data(entry["name"], entry["data"],
                              entry.get("save_copy_and_apply_at_reset", False),
                              entry.get("log_data_across_episode", False))
            else:
                raise Exception("Unknown type of data configure, only support tuple and dictionary")

    def add_pool_for_reset(self, name, data, reset_target):
        """A pool of candidate start values for `reset_target` (data_feed.py:88-105)."""
        self.add_data(name, data, False, False, is_reset_pool=True, reset_target=reset_target)
