// include/kpeg/HuffmanTree.hpp -- Huffman code tree built on the HOST from a DHT segment
// (the north star keeps table construction on the CPU).  Same surface as the reference's
// include/HuffmanTree.hpp: Node/NodePtr, createRootNode/createNode, insertLeft/insertRight,
// getRightLevelNode, inOrder, class HuffmanTree { constructHuffmanTree, getTree, contains }.
//
// contains(code) keeps the reference's contract (src/HuffmanTree.cpp:164-193):
//   ""     no leaf at exactly that bit string,
//   "EOB"  the leaf's symbol is 0x00 (in DC trees too -- quirk Q1 hangs on this),
//   else   the symbol as a decimal string.
// The device look-up tables are derived from the same HuffmanTable; tests check that the
// tree and the tables assign identical (canonical) codes.
#ifndef KPEG_HUFFMAN_TREE_HPP
#define KPEG_HUFFMAN_TREE_HPP

#include <memory>
#include <string>

#include "Types.hpp"

namespace kpeg
{
    struct Node
    {
        Node() : root{ false }, leaf{ false }, code{ "" }, value{ 0x00 }, lChild{ nullptr }, rChild{ nullptr }, parent{ nullptr } {}
        Node( const std::string _code, const UInt16 _val ) :
            root{ false }, leaf{ false }, code{ _code }, value{ _val }, lChild{ nullptr }, rChild{ nullptr }, parent{ nullptr } {}

        bool root;
        bool leaf;
        std::string code;
        UInt16 value;
        std::shared_ptr<Node> lChild, rChild;
        std::shared_ptr<Node> parent;
    };

    typedef std::shared_ptr<Node> NodePtr;

    inline NodePtr createRootNode( const UInt16 value )
    {
        NodePtr root = std::make_shared<Node>( "", value );
        root->root = true;
        return root;
    }

    inline NodePtr createNode()
    {
        return std::make_shared<Node>();
    }

    void insertLeft( NodePtr node, const UInt16 value );
    void insertRight( NodePtr node, const UInt16 value );
    NodePtr getRightLevelNode( NodePtr node );
    void inOrder( NodePtr node );

    class HuffmanTree
    {
        public:
            HuffmanTree();
            HuffmanTree( const HuffmanTable& htable );
            void constructHuffmanTree( const HuffmanTable& htable );
            const NodePtr getTree() const;
            const std::string contains( const std::string& huffCode );

        private:
            NodePtr root_;
    };
}

#endif
