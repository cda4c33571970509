"""Buffered CSV statistics writer with the behaviour of the reference's (alpha_zero/utils/csv_writer.py:14-80): rows are
dictionaries, the first row fixes the column names, the header is written only into an empty file, the file is opened in
append mode on every flush (a restarted run continues the same file), flush every `buffer_size` rows or `flush_interval` s."""
import csv
import os
import time


class CsvWriter:
    def __init__(self, fname, buffer_size=100, flush_interval=60):
        d = os.path.dirname(fname)
        if d and not os.path.exists(d):
            os.makedirs(d)
        self._fname, self._fieldnames = fname, None
        self._header_written = not self._is_empty()
        self._buffer, self._buffer_size, self._flush_interval = [], buffer_size, flush_interval
        self._last_flush_time = time.time()

    def _is_empty(self):
        if not os.path.exists(self._fname):
            return True
        with open(self._fname, "r", encoding="utf8") as f:
            return len(list(csv.reader(f))) == 0

    def write(self, values):
        if self._fieldnames is None:
            self._fieldnames = list(values.keys())
        self._buffer.append(values)
        if len(self._buffer) >= self._buffer_size or time.time() - self._last_flush_time >= self._flush_interval:
            self._flush()

    def close(self):
        self._flush()

    def _flush(self):
        if not self._buffer:
            return
        with open(self._fname, "a") as f:
            w = csv.DictWriter(f, fieldnames=self._fieldnames)
            if not self._header_written:
                w.writeheader()
                self._header_written = True
            w.writerows(self._buffer)
            self._buffer.clear()
        self._last_flush_time = time.time()
