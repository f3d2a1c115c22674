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
#include <cstring>
#include <string>
#include <vector>

namespace alfa {

struct SeqRecord {
    std::string name, comment, seq, qual;
};

class SeqReader {
   public:
    ~SeqReader() { close(); }
    bool open(const char* path) {
        close();
        f_ = gzopen(path, "r");
        if (!f_) return false;
        gzbuffer(f_, 1 << 20);
        buf_.resize(1 << 22);
        beg_ = end_ = 0;
        eof_ = false;
        last_char_ = 0;
        return true;
    }
    void open_mem(const char* data, size_t n) {
        close();
        buf_.assign(data, data + n);
        beg_ = 0; end_ = n;
        eof_ = true;
        last_char_ = 0;
    }
    void close() {
        if (f_) gzclose(f_);
        f_ = nullptr;
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
    bool fill() {   // true when at least one unread byte is available
        if (beg_ < end_) return true;
        if (eof_) return false;
        int got = gzread(f_, buf_.data(), (unsigned)buf_.size());
        beg_ = 0;
        end_ = got > 0 ? (size_t)got : 0;
        if (end_ < buf_.size()) eof_ = true;
        return end_ > 0;
    }
    int getc_() { return fill() ? (unsigned char)buf_[beg_++] : -1; }
    // ks_getuntil(ks, KS_SEP_SPACE): false only when already at EOF
    bool until_space(std::string& s, int& delim) {
        s.clear();
        delim = 0;
        if (!fill()) return false;
        for (;;) {
            size_t i = beg_;
            while (i < end_ && !isspace((unsigned char)buf_[i])) i++;
            s.append(buf_.data() + beg_, i - beg_);
            if (i < end_) { delim = (unsigned char)buf_[i]; beg_ = i + 1; return true; }
            beg_ = end_;
            if (!fill()) return true;
        }
    }
    // ks_getuntil2(ks, KS_SEP_LINE, str, 0, append=1): false at EOF with nothing left
    bool until_line(std::string& s) {
        if (!fill()) return false;
        for (;;) {
            const char* p = buf_.data() + beg_;
            const char* nl = (const char*)memchr(p, '\n', end_ - beg_);
            if (nl) {
                s.append(p, (size_t)(nl - p));
                beg_ = (size_t)(nl - buf_.data()) + 1;
                break;
            }
            s.append(p, end_ - beg_);
            beg_ = end_;
            if (!fill()) break;
        }
        if (s.size() > 1 && s.back() == '\r') s.pop_back();
        return true;
    }
    gzFile f_ = nullptr;
    std::vector<char> buf_;
    size_t beg_ = 0, end_ = 0;
    bool eof_ = true;
    int last_char_ = 0;
};

}  // namespace alfa
