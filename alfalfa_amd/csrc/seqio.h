// seqio.h -- FASTA/FASTQ record reader (plain or gzip) for the host side.
//
// Behavioural contract = klib kseq as vendored by the reference (src/kseq.h:171-211):
// a record starts at the next '>' or '@'; name = header up to the first whitespace,
// comment = rest of the header line; sequence lines are concatenated until a line
// starting with '>', '@' or '+'; after '+' the quality is read until it is at least as
// long as the sequence.  A trailing '\r' is dropped from header/sequence/quality lines.
// Written from scratch over a whole-file buffer (zlib gzread handles plain files too).
#pragma once
#include <zlib.h>

#include <cctype>
#include <string>
#include <vector>

namespace alfa {

struct SeqRecord {
    std::string name, comment, seq, qual;
};

class SeqReader {
   public:
    bool open(const char* path) {
        gzFile f = gzopen(path, "r");
        if (!f) return false;
        buf_.clear();
        std::vector<char> chunk(1 << 20);
        int got;
        while ((got = gzread(f, chunk.data(), (unsigned)chunk.size())) > 0) buf_.insert(buf_.end(), chunk.data(), chunk.data() + got);
        gzclose(f);
        pos_ = 0;
        last_char_ = 0;
        return true;
    }
    void open_mem(const char* data, size_t n) {
        buf_.assign(data, data + n);
        pos_ = 0;
        last_char_ = 0;
    }
    // returns sequence length, -1 at EOF, -2 on truncated quality
    long next(SeqRecord& r) {
        int c;
        if (last_char_ == 0) {
            while ((c = getc_()) != -1 && c != '>' && c != '@') {}
            if (c == -1) return -1;
            last_char_ = c;
        }
        r.comment.clear(); r.seq.clear(); r.qual.clear();
        int delim = 0;
        if (!until_space(r.name, delim)) return -1;
        if (delim != '\n') { r.comment.clear(); until_line(r.comment); }
        while ((c = getc_()) != -1 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            r.seq.push_back((char)c);
            until_line(r.seq);
        }
        if (c == '>' || c == '@') last_char_ = c;
        if (c != '+') return (long)r.seq.size();
        while ((c = getc_()) != -1 && c != '\n') {}
        if (c == -1) return -2;
        while (until_line(r.qual) && r.qual.size() < r.seq.size()) {}
        last_char_ = 0;
        if (r.seq.size() != r.qual.size()) return -2;
        return (long)r.seq.size();
    }

   private:
    int getc_() { return pos_ < buf_.size() ? (unsigned char)buf_[pos_++] : -1; }
    // ks_getuntil(ks, KS_SEP_SPACE): false only when already at EOF
    bool until_space(std::string& s, int& delim) {
        s.clear();
        delim = 0;
        if (pos_ >= buf_.size()) return false;
        size_t i = pos_;
        while (i < buf_.size() && !isspace((unsigned char)buf_[i])) i++;
        s.assign(buf_.data() + pos_, i - pos_);
        if (i < buf_.size()) delim = (unsigned char)buf_[i];
        pos_ = i + 1;
        return true;
    }
    // ks_getuntil2(ks, KS_SEP_LINE, str, 0, append=1): false at EOF with nothing left
    bool until_line(std::string& s) {
        if (pos_ >= buf_.size()) return false;
        size_t i = pos_;
        while (i < buf_.size() && buf_[i] != '\n') i++;
        s.append(buf_.data() + pos_, i - pos_);
        pos_ = i + 1;
        if (s.size() > 1 && s.back() == '\r') s.pop_back();
        return true;
    }
    std::vector<char> buf_;
    size_t pos_ = 0;
    int last_char_ = 0;
};

}  // namespace alfa
